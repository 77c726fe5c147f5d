"""The one-launch sub-block kernels of the small geometry (embedding width <= 256) against the launch-per-step form of
the same C-ABI block calls, and against an fp64 evaluation of the sub-block (mymodel/ViT.py:52-73, 119-120)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def _ff_block(rows, D, M, fused, erf, seed=0):
    from visiontransformer_amd._lib import check, lib
    from visiontransformer_amd.ops import _p, _stream
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, D, generator=g).to(DEV, BF)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    w1 = (torch.randn(M, D, generator=g) * D ** -0.5).to(DEV, BF)
    b1 = (0.1 * torch.randn(M, generator=g)).to(DEV)
    w2 = (torch.randn(D, M, generator=g) * M ** -0.5).to(DEV, BF)
    b2 = (0.1 * torch.randn(D, generator=g)).to(DEV)
    out = dict(xn=torch.empty(rows, D, dtype=BF, device=DEV), mean=torch.empty(rows, device=DEV),
               rstd=torch.empty(rows, device=DEV), u=torch.empty(rows, M, dtype=BF, device=DEV),
               h=torch.empty(rows, M, dtype=BF, device=DEV), y=torch.empty(rows, D, dtype=BF, device=DEV))
    check(lib.vitmi_debug_set_small_fused(-1 if fused else 0))
    check(lib.vitmi_set_gelu_mode(1 if erf else 0))
    try:
        check(lib.vitmi_ff_block_fwd(1, rows, D, M, _p(x), _p(gamma), _p(beta), 1e-5, _p(w1), _p(b1), _p(w2), _p(b2),
                                     _p(out["xn"]), _p(out["mean"]), _p(out["rstd"]), _p(out["u"]), _p(out["h"]),
                                     _p(out["y"]), _stream()), "vitmi_ff_block_fwd")
        torch.cuda.synchronize()
    finally:
        check(lib.vitmi_debug_set_small_fused(-1))
        check(lib.vitmi_set_gelu_mode(0))
    ins = dict(x=x, gamma=gamma, beta=beta, w1=w1, b1=b1, w2=w2, b2=b2)
    return ins, out


@pytest.mark.parametrize("erf", [False, True])
@pytest.mark.parametrize("rows,D,M", [(256 * 65, 192, 512), (64 * 65, 192, 512), (37, 64, 128), (1000, 256, 1024),
                                      (16 * 5 * 3 + 1, 128, 96), (96, 96, 32)])
def test_ff_block_forward_one_launch_against_steps_and_fp64(rows, D, M, erf):
    ins, fused = _ff_block(rows, D, M, True, erf)
    _, steps = _ff_block(rows, D, M, False, erf)
    # the two forms: same arithmetic per element up to the summation order inside the matrix instructions
    for k in ("mean", "rstd"):
        assert rel_err(fused[k], steps[k]) < 1e-6, k
    for k in ("xn", "h", "y"):
        assert rel_err(fused[k].float(), steps[k].float()) < 8e-3, k     # one bf16 unit of the largest element
    uf = (fused["u"].view(torch.float16) if not erf else fused["u"]).float()
    us = (steps["u"].view(torch.float16) if not erf else steps["u"]).float()
    assert rel_err(uf, us) < 8e-3
    # fp64 evaluation from the same bf16 operands
    x = ins["x"].double()
    xn = torch.nn.functional.layer_norm(x, (D,), ins["gamma"].double(), ins["beta"].double(), 1e-5)
    pre = xn.to(BF).double() @ ins["w1"].double().t() + ins["b1"].double()
    h = torch.nn.functional.gelu(pre)
    y = h.to(BF).double() @ ins["w2"].double().t() + ins["b2"].double() + x
    assert rel_err(fused["xn"].double(), xn) < 2e-2
    assert rel_err(fused["h"].double(), h) < 2e-2
    assert rel_err(fused["y"].double(), y) < 2e-2
    if erf:
        assert rel_err(uf.double(), pre) < 2e-2
    else:
        cdf = 0.5 * (1 + torch.erf(pre / 2 ** 0.5))
        assert rel_err(uf.double(), cdf + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5) < 2e-2


def _cifar_step(overlap, fused, seed=3):
    import visiontransformer_amd as vta
    from visiontransformer_amd import ViT
    from visiontransformer_amd._lib import check, lib
    torch.manual_seed(seed)
    m = ViT.VIT(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=3, heads=12, mlp_dim=512).to(DEV).train()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(16, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (16,), generator=g).to(DEV)
    check(lib.vitmi_debug_set_wgrad_overlap(overlap))
    check(lib.vitmi_debug_set_small_fused(int(fused) if fused is not True else -1))
    try:
        logits = m(x)
        torch.nn.functional.cross_entropy(logits, y).backward()
        torch.cuda.synchronize()
    finally:
        check(lib.vitmi_debug_set_wgrad_overlap(-1))
        check(lib.vitmi_debug_set_small_fused(-1))
    return logits.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}


def test_weight_gradients_on_the_second_stream_change_nothing():
    """The same kernels in the same order per stream: logits and every gradient bit-identical with the weight-gradient
    pair on the library's second stream (forced on), on the caller's stream (off) and in the automatic mode."""
    ref_logits, ref = _cifar_step(0, True)
    for mode in (1, -1):
        logits, grads = _cifar_step(mode, True)
        assert torch.equal(logits, ref_logits)
        for k in ref:
            assert torch.equal(grads[k], ref[k]), (mode, k)


@pytest.mark.parametrize("mask", [3, 7])
def test_model_step_with_and_without_the_one_launch_kernels(mask):
    ref_logits, ref = _cifar_step(-1, 0)
    logits, grads = _cifar_step(-1, mask)
    assert rel_err(logits, ref_logits) < 2e-2
    for k in ref:
        assert rel_err(grads[k], ref[k]) < 2e-2, k


def _attn_block(B, N, D, H, DH, fused, qkv_bias, seed=0):
    from visiontransformer_amd._lib import check, lib
    from visiontransformer_amd.ops import _p, _stream
    I = H * DH
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B * N, D, generator=g).to(DEV, BF)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    wq = (torch.randn(3 * I, D, generator=g) * D ** -0.5).to(DEV, BF)
    bq = (0.1 * torch.randn(3 * I, generator=g)).to(DEV) if qkv_bias else None
    wo = (torch.randn(D, I, generator=g) * I ** -0.5).to(DEV, BF)
    bo = (0.1 * torch.randn(D, generator=g)).to(DEV)
    out = dict(xn=torch.empty(B * N, D, dtype=BF, device=DEV), mean=torch.empty(B * N, device=DEV),
               rstd=torch.empty(B * N, device=DEV), qkv=torch.empty(B * N, 3 * I, dtype=BF, device=DEV),
               o=torch.empty(B * N, I, dtype=BF, device=DEV), lse=torch.empty(B, H, N, device=DEV),
               y=torch.empty(B * N, D, dtype=BF, device=DEV))
    check(lib.vitmi_debug_set_small_fused(-1 if fused else 0))
    try:
        check(lib.vitmi_attn_block_fwd(1, B, N, D, H, DH, _p(x), _p(gamma), _p(beta), 1e-5, _p(wq), _p(bq), _p(wo), _p(bo),
                                       _p(out["xn"]), _p(out["mean"]), _p(out["rstd"]), _p(out["qkv"]), _p(out["o"]),
                                       _p(out["lse"]), None, _p(out["y"]), 0, _stream()), "vitmi_attn_block_fwd")
        torch.cuda.synchronize()
    finally:
        check(lib.vitmi_debug_set_small_fused(-1))
    return dict(x=x, gamma=gamma, beta=beta, wq=wq, bq=bq, wo=wo, bo=bo), out


@pytest.mark.parametrize("qkv_bias", [False, True])
@pytest.mark.parametrize("B,N,D,H,DH", [(256, 65, 192, 12, 8), (4, 33, 192, 12, 8), (3, 17, 64, 4, 8), (2, 96, 256, 16, 16),
                                        (2, 50, 128, 4, 16), (5, 1, 64, 4, 8), (2, 81, 96, 4, 8)])
def test_attention_block_forward_one_launch_against_steps_and_fp64(B, N, D, H, DH, qkv_bias):
    ins, fused = _attn_block(B, N, D, H, DH, True, qkv_bias)
    _, steps = _attn_block(B, N, D, H, DH, False, qkv_bias)
    for k in ("mean", "rstd"):
        assert rel_err(fused[k], steps[k]) < 1e-6, k
    for k in ("xn", "qkv", "o", "y"):
        assert rel_err(fused[k].float(), steps[k].float()) < 8e-3, k
    assert rel_err(fused["lse"], steps["lse"]) < 2e-2
    # fp64 evaluation from the same bf16 operands (rounded where the kernels round)
    I = H * DH
    x = ins["x"].double()
    xn = torch.nn.functional.layer_norm(x, (D,), ins["gamma"].double(), ins["beta"].double(), 1e-5)
    qkv = xn.to(BF).double() @ ins["wq"].double().t() + (ins["bq"].double() if qkv_bias else 0)
    q_, k_, v_ = (t.reshape(B, N, H, DH).permute(0, 2, 1, 3) for t in qkv.to(BF).double().split(I, dim=-1))
    dots = (q_ @ k_.transpose(-1, -2)) * DH ** -0.5
    o = (torch.softmax(dots, -1) @ v_).permute(0, 2, 1, 3).reshape(B * N, I)
    y = o.to(BF).double() @ ins["wo"].double().t() + ins["bo"].double() + x
    assert rel_err(fused["qkv"].double(), qkv) < 2e-2
    assert rel_err(fused["o"].double(), o) < 2e-2
    assert rel_err(fused["lse"].double(), torch.logsumexp(dots, -1)) < 2e-2
    assert rel_err(fused["y"].double(), y) < 2e-2


def _ff_block_bwd(rows, D, M, fused, erf, seed=0):
    from visiontransformer_amd._lib import check, lib
    from visiontransformer_amd.ops import _p, _stream
    ins, fwd = _ff_block(rows, D, M, False, erf, seed)          # the saved tensors of the launch-per-step forward
    g = torch.Generator().manual_seed(seed + 100)
    dy = (torch.randn(rows, D, generator=g) * 0.1).to(DEV, BF)
    w1_t, w2_t = ins["w1"].t().contiguous(), ins["w2"].t().contiguous()
    out = dict(dx=torch.empty(rows, D, dtype=BF, device=DEV), dgamma=torch.empty(D, device=DEV),
               dbeta=torch.empty(D, device=DEV), dw1=torch.empty(M, D, device=DEV), db1=torch.empty(M, device=DEV),
               dw2=torch.empty(D, M, device=DEV), db2=torch.empty(D, device=DEV))
    ws = torch.empty(int(lib.vitmi_ff_block_bwd_workspace_bytes(1, rows, D, M)) + 256, dtype=torch.uint8, device=DEV)
    check(lib.vitmi_debug_set_small_fused(7 if fused else 0))      # bit 2: the backward kernel is opt-in
    check(lib.vitmi_set_gelu_mode(1 if erf else 0))
    try:
        check(lib.vitmi_ff_block_bwd(1, rows, D, M, _p(dy), _p(ins["x"]), _p(fwd["xn"]), _p(fwd["u"]), _p(fwd["h"]),
                                     _p(ins["gamma"]), _p(fwd["mean"]), _p(fwd["rstd"]), _p(ins["w1"]), _p(w1_t),
                                     _p(ins["w2"]), _p(w2_t), _p(out["dx"]), _p(out["dgamma"]), _p(out["dbeta"]),
                                     _p(out["dw1"]), _p(out["db1"]), _p(out["dw2"]), _p(out["db2"]), _p(ws), _stream()),
              "vitmi_ff_block_bwd")
        torch.cuda.synchronize()
    finally:
        check(lib.vitmi_debug_set_small_fused(-1))
        check(lib.vitmi_set_gelu_mode(0))
    return ins, fwd, dy, out


@pytest.mark.parametrize("erf", [False, True])
@pytest.mark.parametrize("rows,D,M", [(256 * 65, 192, 512), (64 * 65, 192, 512), (37, 64, 128), (1000, 256, 1024),
                                      (16 * 5 * 3 + 1, 128, 96), (96, 96, 32)])
def test_ff_block_backward_one_launch_against_steps_and_fp64(rows, D, M, erf):
    ins, fwd, dy, fused = _ff_block_bwd(rows, D, M, True, erf)
    _, _, _, steps = _ff_block_bwd(rows, D, M, False, erf)
    # against the launch-per-step form: dxn stays fp32 here instead of being rounded to bf16 on its way to the
    # LayerNorm backward, so the two differ by that rounding
    for k in ("dx", "dgamma", "dbeta", "dw1", "db1", "dw2", "db2"):
        assert rel_err(fused[k].float(), steps[k].float()) < 1e-2, k
    # fp64 evaluation through autograd from the same bf16 operands
    x = ins["x"].double().requires_grad_(True)
    gamma, beta = ins["gamma"].double().requires_grad_(True), ins["beta"].double().requires_grad_(True)
    w1, b1 = ins["w1"].double().requires_grad_(True), ins["b1"].double().requires_grad_(True)
    w2, b2 = ins["w2"].double().requires_grad_(True), ins["b2"].double().requires_grad_(True)
    xn = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
    y = torch.nn.functional.gelu(xn @ w1.t() + b1) @ w2.t() + b2 + x
    y.backward(dy.double())
    ref = dict(dx=x.grad, dgamma=gamma.grad, dbeta=beta.grad, dw1=w1.grad, db1=b1.grad, dw2=w2.grad, db2=b2.grad)
    for k, v in ref.items():
        assert rel_err(fused[k].double(), v) < 2e-2, k


def _attn_block_bwd(B, N, D, H, DH, fused, seed=0):
    """vitmi_attn_block_bwd in the form that leaves the weight gradients to the caller (both null), from the saved
    tensors of the launch-per-step forward; dqkv is read back from its place in the workspace."""
    from visiontransformer_amd._lib import check, lib
    from visiontransformer_amd.ops import _p, _stream
    I = H * DH
    ins, fwd = _attn_block(B, N, D, H, DH, False, False, seed)
    g = torch.Generator().manual_seed(seed + 100)
    dy = (torch.randn(B * N, D, generator=g) * 0.1).to(DEV, BF)
    wq_t, wo_t = ins["wq"].t().contiguous(), ins["wo"].t().contiguous()
    out = dict(dx=torch.empty(B * N, D, dtype=BF, device=DEV), dgamma=torch.empty(D, device=DEV),
               dbeta=torch.empty(D, device=DEV), db_out=torch.empty(D, device=DEV))
    ws = torch.zeros(int(lib.vitmi_attn_block_bwd_workspace_bytes(1, B, N, D, H, DH)) + 256, dtype=torch.uint8, device=DEV)
    check(lib.vitmi_debug_set_small_fused(27 if fused else 11))
    try:
        check(lib.vitmi_attn_block_bwd(1, B, N, D, H, DH, _p(dy), _p(ins["x"]), _p(fwd["xn"]), _p(fwd["qkv"]), _p(fwd["o"]),
                                       _p(fwd["lse"]), _p(ins["gamma"]), _p(fwd["mean"]), _p(fwd["rstd"]), _p(ins["wq"]),
                                       _p(wq_t), _p(ins["wo"]), _p(wo_t), _p(out["dx"]), _p(out["dgamma"]), _p(out["dbeta"]),
                                       None, None, None, _p(out["db_out"]), _p(ws), 0, _stream()), "vitmi_attn_block_bwd")
        torch.cuda.synchronize()
    finally:
        check(lib.vitmi_debug_set_small_fused(-1))
    off = int(lib.vitmi_attn_block_bwd_dqkv_offset(1, B, N, D, H, DH))
    out["dqkv"] = ws[off:off + B * N * 3 * I * 2].view(BF).reshape(B * N, 3 * I).clone()
    return ins, fwd, dy, out


@pytest.mark.parametrize("B,N,D,H,DH", [(256, 65, 192, 12, 8), (4, 33, 192, 12, 8), (3, 17, 64, 4, 8), (2, 96, 256, 16, 16),
                                        (2, 50, 128, 4, 16), (5, 1, 64, 4, 8), (2, 81, 96, 4, 8)])
def test_attention_block_backward_one_launch_against_steps_and_fp64(B, N, D, H, DH):
    ins, fwd, dy, fused = _attn_block_bwd(B, N, D, H, DH, True)
    _, _, _, steps = _attn_block_bwd(B, N, D, H, DH, False)
    # against the launch-per-step form: dO and dxn are not rounded to bf16 on their way through HBM here
    for k in ("dx", "dgamma", "dbeta", "db_out", "dqkv"):
        assert rel_err(fused[k].float(), steps[k].float()) < 1e-2, k
    # fp64 evaluation through autograd from the same bf16 operands
    I = H * DH
    x = ins["x"].double().requires_grad_(True)
    gamma, beta = ins["gamma"].double().requires_grad_(True), ins["beta"].double().requires_grad_(True)
    wq, wo, bo = ins["wq"].double(), ins["wo"].double(), ins["bo"].double().requires_grad_(True)
    xn = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
    qkv = xn @ wq.t()
    qkv.retain_grad()
    q_, k_, v_ = (t.reshape(B, N, H, DH).permute(0, 2, 1, 3) for t in qkv.split(I, dim=-1))
    o = (torch.softmax((q_ @ k_.transpose(-1, -2)) * DH ** -0.5, -1) @ v_).permute(0, 2, 1, 3).reshape(B * N, I)
    y = o @ wo.t() + bo + x
    y.backward(dy.double())
    ref = dict(dx=x.grad, dgamma=gamma.grad, dbeta=beta.grad, db_out=bo.grad, dqkv=qkv.grad)
    for k, v in ref.items():
        assert rel_err(fused[k].double(), v) < 2e-2, k


def test_wgrad_multi_against_single_products():
    """vitmi_gemm_wgrad_multi: 20 jobs over one token axis (two launches of up to 16), ragged 128 x 128 tiles, with the
    slab workspace and without it, against the fp32 products of the bf16 operands and the one-product entry point."""
    import ctypes
    from visiontransformer_amd import ops
    from visiontransformer_amd._lib import check, lib
    T = 16640
    gen = torch.Generator().manual_seed(5)
    shapes = [(288, 192), (192, 96), (512, 192), (192, 512), (8, 8), (136, 264)]
    us, vs, outs, refs = [], [], [], []
    for k in range(20):
        P, Q = shapes[k % len(shapes)]
        us.append(torch.randn(T, P, generator=gen).to(DEV).bfloat16())
        vs.append(torch.randn(T, Q, generator=gen).to(DEV).bfloat16())
        refs.append(us[-1].float().t() @ vs[-1].float())
    for with_ws in (True, False):
        outs = [torch.full((u.shape[1], v.shape[1]), float("nan"), device=DEV) for u, v in zip(us, vs)]
        jobs = (ops._WgradJob * 20)()
        for j, u, v, c in zip(jobs, us, vs, outs):
            j.U, j.ldu, j.V, j.ldv, j.C, j.ldc, j.P, j.Q = u.data_ptr(), u.shape[1], v.data_ptr(), v.shape[1], c.data_ptr(), \
                c.shape[1], u.shape[1], v.shape[1]
        ws = None
        if with_ws:
            nbytes = lib.vitmi_gemm_wgrad_multi_workspace_bytes(ctypes.addressof(jobs), 20, T)
            assert nbytes > 0
            ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        check(lib.vitmi_gemm_wgrad_multi(ctypes.addressof(jobs), 20, T, None if ws is None else ws.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "vitmi_gemm_wgrad_multi")
        for u, v, c, r in zip(us, vs, outs, refs):
            assert rel_err(c, r) < 1e-5
            if u.shape[1] % 8 == 0:
                assert rel_err(c, ops.linear_bwd_w(u, v)) < 1e-5


def test_deferred_weight_gradients_match_immediate_ones():
    """The weight gradients of the small geometry gathered after the backward pass (one vitmi_gemm_wgrad_multi call
    from an autograd-engine callback) against the ones each sub-block launches itself: same values to summation order;
    the queue is not used when a parameter already holds a gradient (accumulation) or carries a hook, and
    torch.autograd.grad gets filled tensors as well."""
    from visiontransformer_amd import ViT, ops
    from visiontransformer_amd._lib import lib
    kw = dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=3, heads=12, mlp_dim=512, dim_head=8)
    torch.manual_seed(0)
    m = ViT.VIT(**kw).to(DEV).set_compute_dtype(torch.bfloat16).train()
    x = torch.randn(32, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (32,), device=DEV)

    def grads(defer, accumulate=False):
        ops.set_defer_wgrad(defer)
        for p in m.parameters():
            p.grad = torch.zeros_like(p) if accumulate else None
        before = dict(ops.deferred_stats)
        torch.nn.functional.cross_entropy(m(x), y).backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters()}, ops.deferred_stats["jobs"] - before["jobs"]

    try:
        # with the deferred products the MLP sub-block's backward is one launch by default (nothing runs beside it);
        # that form against the launch-per-step one is a bf16-bar comparison ...
        ref, n0 = grads(False)
        got, n1 = grads(True)
        assert n0 == 0 and n1 == 4 * kw["depth"]
        for k in ref:
            assert rel_err(got[k], ref[k]) < 2e-2, k
        # ... and with the same sub-block kernels on both sides only the summation order of the products differs
        lib.vitmi_debug_set_small_fused(3)
        ref, n0 = grads(False)
        got, n1 = grads(True)
        assert n0 == 0 and n1 == 4 * kw["depth"]
        for k in ref:
            assert rel_err(got[k], ref[k]) < 1e-5, k
        acc, n2 = grads(True, accumulate=True)          # .grad present: the sub-blocks launch their own products
        assert n2 == 0
        for k in ref:
            assert rel_err(acc[k], ref[k]) < 1e-5, k
        w = dict(m.named_parameters())["transformer.layers.0.1.fn.net.0.weight"]
        params = [p for p in m.parameters()]
        for p in params:
            p.grad = None
        ops.set_defer_wgrad(True)
        gs = torch.autograd.grad(torch.nn.functional.cross_entropy(m(x), y), params)
        for (k, _), g in zip(m.named_parameters(), gs):
            assert rel_err(g, ref[k]) < 1e-5, k
        # a pass started from inside another one (reentrant checkpointing) has a queue of its own
        from torch.utils.checkpoint import checkpoint
        for p in params:
            p.grad = None
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.cross_entropy(checkpoint(m, xr, use_reentrant=True), y).backward()
        torch.cuda.synchronize()
        for k, p in m.named_parameters():              # (the checkpointed forward runs the no-grad entry points:
            assert rel_err(p.grad, ref[k]) < 2e-2, k   # the loss gradient differs at the bf16 level)
        # a pass that wants the input gradient only (torch.autograd.grad(loss, x), as an adversarial-example loop does):
        # autograd drops the parameter gradients the nodes hand back, so the queue must not write into their memory
        for p in params:
            p.grad = None
        xg = x.clone().requires_grad_(True)
        dropped0 = ops.deferred_stats.get("dropped", 0)
        (dx1,) = torch.autograd.grad(torch.nn.functional.cross_entropy(m(xg), y), [xg])
        torch.cuda.synchronize()
        n_dropped = ops.deferred_stats.get("dropped", 0) - dropped0
        assert n_dropped >= 4 * kw["depth"], n_dropped
        assert all(p.grad is None for p in params)
        ops.set_defer_wgrad(False)
        (dx0,) = torch.autograd.grad(torch.nn.functional.cross_entropy(m(xg), y), [xg])
        ops.set_defer_wgrad(True)
        assert rel_err(dx1, dx0) < 1e-5
        h = w.register_hook(lambda g: g)                # a tensor hook on one weight: its sub-block does not defer
        hooked, n3 = grads(True)
        h.remove()
        assert n3 == 4 * kw["depth"] - 2
        for k in ref:
            assert rel_err(hooked[k], ref[k]) < 1e-5, k
    finally:
        ops.set_defer_wgrad(True)
        lib.vitmi_debug_set_small_fused(-1)


def test_deferred_vector_sums_match_immediate_ones():
    """The bias / LayerNorm-parameter gradients of the one-launch sub-block kernels reduced by one
    vitmi_small_sums_multi launch after the backward pass against the reduction each sub-block call runs itself: the
    same partial sums through the same arithmetic, so bit for bit; not used when a vector parameter already holds a
    gradient or carries a hook; a call that asks for it on a shape without the one-launch kernel is refused."""
    from visiontransformer_amd import ViT, ops
    from visiontransformer_amd._lib import lib
    kw = dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=3, heads=12, mlp_dim=512, dim_head=8)
    torch.manual_seed(0)
    m = ViT.VIT(**kw).to(DEV).set_compute_dtype(torch.bfloat16).train()
    x = torch.randn(32, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (32,), device=DEV)

    def grads(defer_sums, accumulate=False):
        ops.set_defer_sums(defer_sums)
        for p in m.parameters():
            p.grad = torch.zeros_like(p) if accumulate else None
        before = ops.deferred_stats.get("sums", 0)
        torch.nn.functional.cross_entropy(m(x), y).backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters()}, ops.deferred_stats.get("sums", 0) - before

    try:
        ref, n0 = grads(False)
        got, n1 = grads(True)
        assert n0 == 0 and n1 == 2 * kw["depth"]
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
        acc, n2 = grads(True, accumulate=True)          # .grad present: every sub-block forms its own gradients, and
        assert n2 == 0                                  # through the launch-per-step backward: a bf16-bar comparison
        for k in ref:
            assert rel_err(acc[k], ref[k]) < 2e-2, k
        params = list(m.parameters())
        for p in params:
            p.grad = None
        gs = torch.autograd.grad(torch.nn.functional.cross_entropy(m(x), y), params)
        torch.cuda.synchronize()
        for (k, _), g in zip(m.named_parameters(), gs):
            assert torch.equal(g, ref[k]), k
        ln = dict(m.named_parameters())["transformer.layers.1.1.norm.weight"]
        h = ln.register_hook(lambda g: g)               # a tensor hook on one LayerNorm weight: its sub-block does not defer
        hooked, n3 = grads(True)
        h.remove()
        assert n3 == 2 * kw["depth"] - 1
        for k in ref:
            assert torch.equal(hooked[k], ref[k]), k
        # more sub-blocks than one launch takes (32): a narrow model of 17 layers
        kw2 = dict(image_size=32, patch_size=4, embed_dim=64, num_classes=10, depth=17, heads=4, mlp_dim=64, dim_head=16)
        m2 = ViT.VIT(**kw2).to(DEV).set_compute_dtype(torch.bfloat16).train()

        def grads2(defer_sums):
            ops.set_defer_sums(defer_sums)
            for p in m2.parameters():
                p.grad = None
            before = ops.deferred_stats.get("sums", 0)
            torch.nn.functional.cross_entropy(m2(x), y).backward()
            torch.cuda.synchronize()
            return {k: p.grad.clone() for k, p in m2.named_parameters()}, ops.deferred_stats.get("sums", 0) - before

        ref2, n4 = grads2(False)
        got2, n5 = grads2(True)
        assert n4 == 0 and n5 == 2 * kw2["depth"]
        for k in ref2:
            assert torch.equal(got2[k], ref2[k]), k
        # the C entry refuses the request where no one-launch kernel exists (ViT-B/16 widths)
        assert lib.vitmi_ff_block_bwd_sums_strips(1, 50432, 768, 3072) == 0
        assert lib.vitmi_attn_block_bwd_sums_strips(1, 256, 197, 768, 12, 64) == 0
        assert lib.vitmi_attn_block_bwd_sums_strips(1, 32, 65, 192, 12, 8) == 32
    finally:
        ops.set_defer_sums(True)
