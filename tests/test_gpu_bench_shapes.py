"""GPU parity at the shapes bench.py runs: ViT-B/16 224 px at batch 256 has M = 256 * 197 = 50432 token rows, and
every encoder GEMM of that step has at least 256 output tiles of 256 x 256, so `pick_nt_cfg` sends it to the
persistent `gemm_nt_phased_kernel` (K % 64 == 0: 12, 36 or 48 K-tiles, 48-192 streamed regions, every block
walking 2-10 tiles) and every weight gradient to `gemm_tn_phased_kernel` + slab reduction.  The small-shape tests
in test_gpu_kernels.py reach those kernels with at most three K-tiles; here they run at depth.

Reference: the fp32 CPU product of the bf16-rounded operands (every output element, not a sample), with the
epilogue evaluated by the oracle's fp32 expressions (mymodel/ViT.py:66-70, 86-89, 119-120).  Bar: 1e-2 relative
(max|a-b| / max|b|), tighter than the 2e-2 bf16 bar of the model tests; fp32 weight gradients 1e-4.
The whole-model case runs one ViT-B/16 block at batch 128 (25216 rows, ragged last m-tile, 297-1188 tiles per GEMM:
the same kernels) against `vit_oracle.loss_and_grads` on the CPU.
"""
import pytest
import torch

from conftest import rel_err
from oracle import vit_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
M_BENCH = 256 * 197


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t):
    return t.to(BF).float()


def deriv(t):
    """the derivative buffer of the GELU_D / MULAUX pair as fp32: IEEE half bits in a bf16-typed tensor, plain fp32
    in fp32 mode"""
    return t.view(torch.float16).float() if t.dtype == torch.bfloat16 else t.float()


@pytest.fixture(scope="module")
def ops():
    from visiontransformer_amd import ops as o
    return o


# (N, K) of the five distinct nt products of a ViT-B/16 block: qkv, out projection, fc2, dX of qkv, fc1;
# the ragged M leaves 100 rows of the last 256-row tile empty
@pytest.mark.parametrize("M,N,K", [(M_BENCH, 2304, 768), (M_BENCH, 768, 768), (M_BENCH, 768, 3072),
                                   (M_BENCH, 768, 2304), (M_BENCH, 3072, 768), (M_BENCH - 100, 768, 768),
                                   (M_BENCH - 100, 3072, 768),
                                   # the 384-wide model (timm-style ViT-S/16): N = 384 and 1152 end in a half-empty 256-wide column tile,
                                   # K = 384 is six K-tiles, 64 a single one
                                   (M_BENCH, 384, 384), (M_BENCH, 1152, 384), (M_BENCH, 384, 1536), (M_BENCH, 384, 1152),
                                   (M_BENCH - 100, 384, 384), (M_BENCH - 3, 1152, 64), (M_BENCH, 384, 128)])
def test_phased_nt_gemm_forward_epilogues_at_bench_shapes(ops, M, N, K):
    """linear_fwd -> gemm_nt_phased_kernel<NONE,bf16> (plain, bias + residual), <GELU,bf16> (bias, pre-activation
    written out) and <NONE,f32>."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    xd, wd, bd = x.to(DEV, BF), w.to(DEV, BF), b.to(DEV)
    pre = q(x) @ q(w).t()
    y, _ = ops.linear_fwd(xd, wd)
    assert rel_err(y, pre) < 1e-2
    res = rnd(M, N, seed=4)
    y, _ = ops.linear_fwd(xd, wd, bd, res=res.to(DEV, BF))
    assert rel_err(y, pre + b + q(res)) < 1e-2
    del res
    if N == 3072:                                       # the first MLP layer is the only GELU product of the step
        h, u = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_GELU, want_aux=True)
        assert rel_err(u, pre + b) < 1e-2
        assert rel_err(h, vit_oracle.gelu_erf(pre + b)) < 1e-2
        # <GELU_D,bf16>: the form the MLP block runs -- aux receives gelu'(pre) instead of pre
        h2, gp = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_GELU_D, want_aux=True)
        pr = (pre + b).requires_grad_(True)
        vit_oracle.gelu_erf(pr).sum().backward()
        # <GELU> evaluates the erf form, <GELU_D> the fitted sigmoid form (3.8e-5 apart before rounding): equal up to
        # a bf16 unit in the last place
        assert rel_err(h2, h) < 8e-3 and rel_err(deriv(gp), pr.grad) < 1e-2
    if (N == 768 and K == 768) or (N == 384 and K == 384):
        y32, _ = ops.linear_fwd(xd, wd, bd, out_dtype=torch.float32)
        assert rel_err(y32, pre + b) < 2e-3


# (rows, width of dy, width of dx): dX of fc2 (into the dGELU epilogue), of fc1, of qkv, of the out projection
@pytest.mark.parametrize("M,N,K", [(M_BENCH, 768, 3072), (M_BENCH, 3072, 768), (M_BENCH, 2304, 768),
                                   (M_BENCH - 100, 768, 3072)])
def test_phased_nt_gemm_input_gradients_at_bench_shapes(ops, M, N, K):
    """linear_bwd_x through the transposed weight copy -> gemm_nt_phased_kernel<NONE,bf16>; with K = 3072 also
    <DGELU,bf16,false> and the fused column-sum variant <DGELU,bf16,true> (bias gradient of the first MLP layer)."""
    dy, w = rnd(M, N, seed=5), rnd(N, K, seed=6, scale=N ** -0.5)
    dyd = dy.to(DEV, BF)
    wp = torch.nn.Parameter(w.to(DEV))
    wd, wt = ops.compute_copy(wp, BF), ops.compute_copy_t(wp, BF)
    ref = q(dy) @ q(w)
    dx = ops.linear_bwd_x(dyd, wd, w_t=wt)
    assert rel_err(dx, ref) < 1e-2
    if K != 3072:
        return
    aux = rnd(M, K, seed=7)
    auxd = aux.to(DEV, BF)
    a = q(aux).requires_grad_(True)
    vit_oracle.gelu_erf(a).backward(ref)
    du = ops.linear_bwd_x(dyd, wd, epi=ops.EPI_DGELU, aux=auxd, w_t=wt)
    assert rel_err(du, a.grad) < 1e-2
    du2, cs = ops.linear_bwd_x(dyd, wd, epi=ops.EPI_DGELU, aux=auxd, w_t=wt, want_colsum=True)
    assert torch.equal(du2, du)
    assert rel_err(cs, a.grad.double().sum(0)) < 1e-2
    assert rel_err(cs, du.double().sum(0)) < 2e-3          # sums of the unrounded values vs sums of their bf16
    # <MULAUX,bf16,false/true>: the backward half of the pair above (aux holds the derivative itself)
    auxh = aux.half().to(DEV).view(BF)                              # the pair keeps its aux as IEEE half bits
    gref = ref * aux.half().float()
    dm = ops.linear_bwd_x(dyd, wd, epi=ops.EPI_MULAUX, aux=auxh, w_t=wt)
    assert rel_err(dm, gref) < 1e-2
    dm2, csm = ops.linear_bwd_x(dyd, wd, epi=ops.EPI_MULAUX, aux=auxh, w_t=wt, want_colsum=True)
    assert torch.equal(dm2, dm)
    assert rel_err(csm, gref.double().sum(0)) < 1e-4          # fp32 sums of the unrounded products
    assert rel_err(csm, dm.double().sum(0)) < 5e-3            # ... against sums of their bf16 roundings


@pytest.mark.parametrize("T,P,Q", [(M_BENCH, 2304, 768), (M_BENCH, 768, 768), (M_BENCH, 3072, 768),
                                   (M_BENCH, 768, 3072)])
def test_phased_weight_gradient_gemm_at_bench_shapes(ops, T, P, Q):
    """linear_bwd_w -> gemm_tn_phased_kernel over 788 token tiles, split into one round of slabs."""
    dy, x = rnd(T, P, seed=8), rnd(T, Q, seed=9)
    dw = ops.linear_bwd_w(dy.to(DEV, BF), x.to(DEV, BF))
    ref = q(dy).double().t() @ q(x).double()
    assert rel_err(dw, ref) < 1e-4
    assert torch.equal(dw, ops.linear_bwd_w(dy.to(DEV, BF), x.to(DEV, BF)))


@pytest.mark.parametrize("T,Pa,Qa,Pb,Qb", [
    (M_BENCH, 768, 3072, 3072, 768),       # MLP pair: 72 output tiles, three slabs each and a shared remainder
    (M_BENCH, 768, 768, 2304, 768),        # attention pair: 36 tiles, seven equal slabs
    (12608, 768, 3072, 3072, 768),         # 197 token tiles
    (64 * 37, 256, 512, 512, 256),         # 4 tiles, 37 token tiles: slabs capped by the token count
    (64 * 100, 768, 1024, 1280, 768),      # 27 tiles: nine slabs and a remainder over 13 blocks
    (M_BENCH - 100, 768, 768, 2304, 768),  # tokens not a multiple of 64: two single launches
    (6400, 768, 200, 2304, 768),           # an extent that is no multiple of 256: two single launches
])
def test_weight_gradient_pair(ops, T, Pa, Qa, Pb, Qb):
    """linear_bwd_w_pair -> gemm_tn_batch_kernel + tn_batch_reduce_kernel (or two single products) against the fp64
    product of the bf16-rounded operands, bit-identical on a second call and with the one-launch form switched off
    to 1e-6 (other summation order)."""
    from visiontransformer_amd._lib import lib
    dya, xa, dyb, xb = rnd(T, Pa, seed=11), rnd(T, Qa, seed=12), rnd(T, Pb, seed=13), rnd(T, Qb, seed=14)
    args = [t.to(DEV, BF) for t in (dya, xa, dyb, xb)]
    dwa, dwb = ops.linear_bwd_w_pair(*args)
    assert rel_err(dwa, q(dya).double().t() @ q(xa).double()) < 1e-4
    assert rel_err(dwb, q(dyb).double().t() @ q(xb).double()) < 1e-4
    dwa2, dwb2 = ops.linear_bwd_w_pair(*args)
    assert torch.equal(dwa, dwa2) and torch.equal(dwb, dwb2)
    try:
        lib.vitmi_debug_set_tn_batch(0)
        sa, sb = ops.linear_bwd_w_pair(*args)
    finally:
        lib.vitmi_debug_set_tn_batch(1)
    assert torch.equal(sa, ops.linear_bwd_w(args[0], args[1])) and torch.equal(sb, ops.linear_bwd_w(args[2], args[3]))
    assert rel_err(dwa, sa) < 1e-5 and rel_err(dwb, sb) < 1e-5


@pytest.mark.parametrize("reserved", [8, 16, 64])
def test_weight_gradient_pair_with_reserved_compute_units(ops, reserved):
    """vitmi_set_reserved_cus shrinks the grid the pair launch plans for (data-parallel runs leave compute units to the
    collective): other slab counts and remainder runs, same products."""
    from visiontransformer_amd._lib import check, lib
    T = 12608
    dya, xa, dyb, xb = rnd(T, 768, seed=21), rnd(T, 3072, seed=22), rnd(T, 3072, seed=23), rnd(T, 768, seed=24)
    args = [t.to(DEV, BF) for t in (dya, xa, dyb, xb)]
    try:
        check(lib.vitmi_set_reserved_cus(reserved), "vitmi_set_reserved_cus")
        dwa, dwb = ops.linear_bwd_w_pair(*args)
        dwc, dwd = ops.linear_bwd_w_pair(args[0][:, :768], args[0], args[2][:, :2304].contiguous(), args[3])
    finally:
        check(lib.vitmi_set_reserved_cus(0), "vitmi_set_reserved_cus")
    assert rel_err(dwa, q(dya).double().t() @ q(xa).double()) < 1e-4
    assert rel_err(dwb, q(dyb).double().t() @ q(xb).double()) < 1e-4
    assert rel_err(dwc, q(dya).double().t() @ q(dya).double()) < 1e-4
    assert rel_err(dwd, q(dyb[:, :2304]).double().t() @ q(xb).double()) < 1e-4


def test_vit_b16_block_at_batch_128_against_oracle():
    """One ViT-B/16 encoder block (mymodel/ViT.py:93-121 with D 768, H 12, dh 64, mlp 3072) at batch 128, bf16:
    25216 token rows, so qkv / out / fc1 / fc2 and their input gradients have 297-1188 output tiles and run on
    gemm_nt_phased_kernel (all epilogues), the four weight gradients on gemm_tn_phased_kernel, attention on the
    13-tile N = 197 instantiations -- the kernels of the benchmark step.  Logits, loss and every gradient against
    the CPU oracle, 2e-2 relative."""
    from visiontransformer_amd import ViT
    kw = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=1, heads=12, mlp_dim=3072,
              dim_head=64)
    params = vit_oracle.init_params(masked=False, seed=5, **kw)
    g = torch.Generator().manual_seed(17)
    B = 128
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 100, (B,), generator=g)
    logits_ref, loss_ref, grads_ref = vit_oracle.loss_and_grads(params, x, y, 12)
    m = ViT.VIT(**kw)
    m.load_state_dict(params)
    m = m.to(DEV).set_compute_dtype(BF).train()
    logits = m(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, y.to(DEV))
    loss.backward()
    assert rel_err(logits, logits_ref) < 2e-2
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item()))
    worst = max((rel_err(p.grad, grads_ref[k]), k) for k, p in m.named_parameters())
    assert worst[0] < 2e-2, worst


@pytest.mark.parametrize("dtype,tol,B", [(torch.float32, 1e-4, 8), (BF, 2e-2, 32), (BF, 2e-2, 8)])
def test_vit_b16_full_depth_against_oracle(dtype, tol, B):
    """The benchmark's model at its full depth (12 blocks of D 768, H 12, dh 64, mlp 3072, 197 tokens): what twelve
    layers of rounding add up to, against the CPU oracle.  Logits, loss and every gradient, all at the plain bar.

    One documented exception, bf16 at batch 8: a gradient element there is a sum over eight images only and single
    elements of the classifier weight's gradient reach 2.0-2.6e-2 by the maximum measure (rms 0.9e-2).  The cause is
    the bf16 residual stream: torch under bf16 autocast, whose residual stream stays fp32, is at 0.6e-2 on that
    tensor, while the SAME oracle run by torch on this GPU with parameters and every activation in bf16 is at 2.2e-2
    (tools/depth_diag.py prints all three).  A tensor above the bar is therefore accepted only if that all-bf16 run
    is above the bar on it too, the HIP path is within 1.5x of it, and the tensor's rms error is under the bar; at
    most two tensors may take that exit."""
    from visiontransformer_amd import ViT
    kw = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072,
              dim_head=64)
    params = vit_oracle.init_params(masked=False, seed=9, **kw)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 100, (B,), generator=g)
    logits_ref, loss_ref, grads_ref = vit_oracle.loss_and_grads(params, x, y, 12)
    m = ViT.VIT(**kw)
    m.load_state_dict(params)
    m = m.to(DEV).set_compute_dtype(dtype).train()
    logits = m(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, y.to(DEV))
    loss.backward()
    assert rel_err(logits, logits_ref) < tol
    assert abs(loss.item() - loss_ref.item()) < tol * max(1.0, abs(loss_ref.item()))
    errs = sorted(((rel_err(p.grad, grads_ref[k]), k) for k, p in m.named_parameters()), reverse=True)
    over = [(e, k) for e, k in errs if not e < tol]
    if over and dtype == BF and B == 8:
        assert len(over) <= 2, errs[:5]
        pb = {k: v.to(DEV, BF) for k, v in params.items()}
        _, _, grads_bf = vit_oracle.loss_and_grads(pb, x.to(DEV, BF), y.to(DEV), 12)
        grads_hip = dict((k, p.grad) for k, p in m.named_parameters())
        for e, k in over:
            yard = rel_err(grads_bf[k].float(), grads_ref[k])
            ref = grads_ref[k].double()
            rms = ((grads_hip[k].double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            assert yard > tol and e <= 1.5 * yard and rms < tol, (k, e, yard, rms)
    else:
        assert not over, errs[:5]


@pytest.mark.parametrize("T,P0,Q0,P1,Q1", [(50432, 384, 1536, 1536, 384), (50432, 768, 3072, 3072, 768),
                                            (50432, 768, 768, 2304, 768), (12608, 384, 384, 1152, 384)])
def test_weight_gradient_workspace_holds_under_every_cu_reservation(T, P0, Q0, P1, Q1):
    """A workspace sized while a CU reservation was in force (GradReducer attached) must hold after the reservation
    is gone (more slabs / slots per tile): the launches stay inside the queried bytes -- a guard region behind the
    workspace is untouched -- and give the same gradients."""
    from visiontransformer_amd import ops
    from visiontransformer_amd._lib import check, lib
    g = torch.Generator().manual_seed(T + P0)
    dya = (torch.randn(T, P0, generator=g) * 0.1).to(DEV, torch.bfloat16)
    xa = torch.randn(T, Q0, generator=g).to(DEV, torch.bfloat16)
    dyb = (torch.randn(T, P1, generator=g) * 0.1).to(DEV, torch.bfloat16)
    xb = torch.randn(T, Q1, generator=g).to(DEV, torch.bfloat16)
    results = []
    try:
        for size_at, run_at in ((16, 0), (0, 32), (128, 0), (0, 0)):
            check(lib.vitmi_set_reserved_cus(size_at))
            nbytes = int(lib.vitmi_gemm_wgrad_pair_workspace_bytes(1, T, P0, Q0, P1, Q1))
            guard = 1 << 20
            ws = torch.empty(nbytes + guard, dtype=torch.uint8, device=DEV)
            ws[nbytes:] = 0xA5
            check(lib.vitmi_set_reserved_cus(run_at))
            dwa = torch.empty(P0, Q0, device=DEV)
            dwb = torch.empty(P1, Q1, device=DEV)
            p = ops._p
            check(lib.vitmi_gemm_wgrad_pair(1, T, P0, Q0, p(dya), P0, p(xa), Q0, p(dwa), P1, Q1, p(dyb), P1, p(xb), Q1,
                                            p(dwb), p(ws), ops._stream()), "vitmi_gemm_wgrad_pair")
            check(lib.vitmi_gemm_wgrad(1, P0, Q0, T, p(dya), P0, p(xa), Q0, p(dwa), Q0, p(ws), ops._stream()))
            torch.cuda.synchronize()
            assert bool((ws[nbytes:] == 0xA5).all()), (size_at, run_at)
            results.append((dwa.clone(), dwb.clone()))
    finally:
        check(lib.vitmi_set_reserved_cus(0))
    for a, b in results[1:]:
        assert rel_err(a, results[0][0]) < 1e-5 and rel_err(b, results[0][1]) < 1e-5


@pytest.mark.parametrize("B,with_bias", [(128, False), (9, False), (32, True)])
def test_attention_block_planes_equal_rows(B, with_bias):
    """The attention sub-block with qkv / dqkv as head-major planes (the default for bf16 heads of 64) against the same
    calls on the row layout (vitmi_debug_set_qkv_planes(0)): the GEMMs store and load the same values at other
    addresses, so the output and every gradient are equal bit for bit.  B = 9: tokens no multiple of 64 (the paired
    weight-gradient launch does not apply, the planes go through the single products); with_bias: the timm-style qkv
    bias, whose gradient keeps dqkv in rows."""
    from visiontransformer_amd import ops
    from visiontransformer_amd._lib import lib
    N, D, H = 197, 768, 12
    g = torch.Generator().manual_seed(31 + B)
    x = torch.randn(B, N, D, generator=g).to(DEV, torch.bfloat16).requires_grad_(True)
    dy = torch.randn(B, N, D, generator=g).to(DEV, torch.bfloat16)
    ln_w = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV).requires_grad_(True)
    ln_b = (0.1 * torch.randn(D, generator=g)).to(DEV).requires_grad_(True)
    wq = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(DEV).requires_grad_(True)
    bq = (0.1 * torch.randn(3 * D, generator=g)).to(DEV).requires_grad_(True) if with_bias else None
    wo = (torch.randn(D, D, generator=g) * D ** -0.5).to(DEV).requires_grad_(True)
    bo = (0.1 * torch.randn(D, generator=g)).to(DEV).requires_grad_(True)
    leaves = [t for t in (x, ln_w, ln_b, wq, wo, bo, bq) if t is not None]
    res = {}
    try:
        for mask in (3, 1, 0):
            lib.vitmi_debug_set_qkv_planes(mask)
            for t in leaves:
                t.grad = None
            y, _ = ops.AttnBlockFn.apply(x, ln_w, ln_b, wq, wo, bo, H, 1e-5, True, False, torch.bfloat16, bq)
            y.backward(dy)
            res[mask] = [y.detach().clone()] + [t.grad.clone() for t in leaves]
    finally:
        lib.vitmi_debug_set_qkv_planes(3)
    for mask in (3, 1):
        for a, b in zip(res[mask], res[0]):
            assert torch.equal(a, b), mask
