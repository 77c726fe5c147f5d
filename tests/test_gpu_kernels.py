"""GPU parity tests, kernel by kernel: each C-ABI entry point against the CPU oracle (or the
plain fp32 torch expression of the same op) on seeded inputs.  fp32 kernels: 1e-4 relative;
bf16 kernels: 2e-2 relative (relative = max|a-b| / max|b|)."""
import math

import numpy as np

import pytest
import torch

from conftest import rel_err
from oracle import vit_oracle

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def deriv(t):
    """the derivative buffer of the GELU_D / MULAUX pair as fp32: IEEE half bits in a bf16-typed tensor, plain fp32
    in fp32 mode"""
    return t.view(torch.float16).float() if t.dtype == torch.bfloat16 else t.float()


@pytest.fixture(scope="module")
def ops():
    from visiontransformer_amd import ops as o
    return o


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def q(t, dtype):
    """value the kernel really sees: rounded to the compute dtype, back in fp32 on the CPU"""
    return t.detach().to(dtype).float().clone()


# ------------------------------------------------------------------------------------------ GEMM
SHAPES = [(128, 128, 64), (256, 384, 192), (197, 288, 192), (65, 96, 96), (50, 10, 48), (1, 100, 768),
          (300, 200, 72), (130, 130, 130), (64, 512, 1000)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_forward_backward_products(ops, dtype, M, N, K):
    x, w, dy = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(M, N, seed=3)
    bias = rnd(N, seed=4)
    xd, wd, dyd = x.to(DEV, dtype), w.to(DEV, dtype), dy.to(DEV, dtype)
    y, _ = ops.linear_fwd(xd, wd, bias.to(DEV))
    assert rel_err(y, q(x, dtype) @ q(w, dtype).t() + bias) < TOL[dtype]
    dx = ops.linear_bwd_x(dyd, wd)
    assert rel_err(dx, q(dy, dtype) @ q(w, dtype)) < TOL[dtype]
    dw = ops.linear_bwd_w(dyd, xd)
    assert dw.dtype == torch.float32
    assert rel_err(dw, q(dy, dtype).t() @ q(x, dtype)) < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(ops, dtype):
    M, N, K = 200, 136, 96
    x, w, b, r = rnd(M, K, seed=5), rnd(N, K, seed=6, scale=K ** -0.5), rnd(N, seed=7), rnd(M, N, seed=8)
    xd, wd, bd, rd = x.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV), r.to(DEV, dtype)
    pre = q(x, dtype) @ q(w, dtype).t() + b
    y, _ = ops.linear_fwd(xd, wd, bd, res=rd)
    assert rel_err(y, pre + q(r, dtype)) < TOL[dtype]
    h, u = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_GELU, want_aux=True)
    assert rel_err(u, pre) < TOL[dtype]
    assert rel_err(h, vit_oracle.gelu_erf(pre)) < TOL[dtype]
    # dGELU epilogue: (dy @ W2) * gelu'(u)
    w2, dy = rnd(K, N, seed=9, scale=N ** -0.5), rnd(M, K, seed=10)
    du = ops.linear_bwd_x(dy.to(DEV, dtype), w2.to(DEV, dtype), epi=ops.EPI_DGELU, aux=u)
    uu = u.float().cpu().requires_grad_(True)
    vit_oracle.gelu_erf(uu).backward(q(dy, dtype) @ q(w2, dtype))
    assert rel_err(du, uu.grad) < TOL[dtype]
    # fp32 output from bf16 operands with bias (the classifier head)
    y32, _ = ops.linear_fwd(xd, wd, bd, out_dtype=torch.float32)
    assert y32.dtype == torch.float32 and rel_err(y32, pre) < (1e-4 if dtype == torch.float32 else 2e-3)
    # the same pair with the derivative evaluated in the forward epilogue and multiplied in by the backward one
    h2, gp = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_GELU_D, want_aux=True)
    pr = pre.clone().requires_grad_(True)
    vit_oracle.gelu_erf(pr).sum().backward()
    assert rel_err(h2, vit_oracle.gelu_erf(pre)) < TOL[dtype] and rel_err(deriv(gp), pr.grad) < TOL[dtype]
    du2 = ops.linear_bwd_x(dy.to(DEV, dtype), w2.to(DEV, dtype), epi=ops.EPI_MULAUX, aux=gp)
    assert rel_err(du2, (q(dy, dtype) @ q(w2, dtype)) * deriv(gp).cpu()) < TOL[dtype]


def test_gemm_strided_views_and_unaligned(ops):
    # operands that are column slices of wider buffers (row stride > K) and odd sizes force the scalar loaders
    big = rnd(77, 300, seed=11).to(DEV, torch.bfloat16)
    x = big[:, 3:3 + 50]                      # unaligned start, K = 50
    w = rnd(33, 50, seed=12).to(DEV, torch.bfloat16)
    y, _ = ops.linear_fwd(x, w)
    assert rel_err(y, x.float().cpu() @ w.float().cpu().t()) < 2e-2


# ------------------------------------------------------------------------------------------ LayerNorm / reductions
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,D", [(4 * 65, 192), (197, 768), (7, 96), (3, 50), (1030, 384)])
def test_layernorm(ops, dtype, rows, D):
    x, g, b, dy, res = rnd(rows, D, seed=1), 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3), rnd(rows, D, seed=4), \
        rnd(rows, D, seed=5)
    xd = x.to(DEV, dtype)
    y, mean, rstd = ops.layernorm_fwd(xd, g.to(DEV), b.to(DEV), 1e-5)
    xr = q(x, dtype).requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = vit_oracle.layer_norm(xr, gr, br)
    assert rel_err(y, yr) < TOL[dtype]
    assert rel_err(mean, xr.mean(-1)) < 1e-5 + (0 if dtype == torch.float32 else 1e-3)
    yr.backward(q(dy, dtype))
    dx, dg, db, rsum = ops.layernorm_bwd(dy.to(DEV, dtype), xd, g.to(DEV), mean, rstd, dres=res.to(DEV, dtype),
                                         want_dres_colsum=True)
    assert rel_err(dx, xr.grad + q(res, dtype)) < TOL[dtype]
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    assert rel_err(rsum, q(res, dtype).double().sum(0)) < 1e-5       # fused bias-gradient reduction
    dx2, _, _ = ops.layernorm_bwd(dy.to(DEV, dtype), xd, g.to(DEV), mean, rstd)
    assert rel_err(dx2, xr.grad) < TOL[dtype]


def test_layernorm_strided_rows(ops):
    # the head normalises x[:, 0] in place: rows N*D apart
    x = rnd(6, 5, 64, seed=6).to(DEV)
    g, b = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    y, _, _ = ops.layernorm_fwd(x[:, 0], g, b, 1e-5)
    assert rel_err(y, vit_oracle.layer_norm(x[:, 0].cpu(), g.cpu(), b.cpu())) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(1000, 768), (13, 10), (50432, 96), (257, 100)])
def test_colsum(ops, dtype, rows, cols):
    x = rnd(rows, cols, seed=7).to(DEV, dtype)
    assert rel_err(ops.colsum(x), x.double().cpu().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------ tokens + PE
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("masked", [False, True])
def test_tokens_assemble(ops, dtype, masked):
    B, N, D = 5, 17, 96
    patch, cls = rnd(B, N - 1, D, seed=1), rnd(1, 1, D, seed=2)
    pe = rnd(N + 1 if masked else N, D, seed=3)
    seq = None
    if masked:
        m = torch.zeros(1, N, dtype=torch.long)
        m[0, [2, 3, 9, 16]] = 1
        seq = vit_oracle.seq_ord_from_mask(m, N).squeeze(0)
    pr, cr, per = q(patch, dtype).requires_grad_(True), cls.clone().requires_grad_(True), pe.clone().requires_grad_(True)
    ref = torch.cat([cr.expand(B, -1, -1), pr], 1) + (per[seq] if masked else per[:N])
    pd = patch.detach().to(DEV, dtype).requires_grad_(True)
    cd, ped = cls.detach().to(DEV).requires_grad_(True), pe.detach().to(DEV).requires_grad_(True)
    out = ops.TokensFn.apply(pd, cd, ped, seq.to(DEV) if masked else None)
    assert rel_err(out, ref) < TOL[dtype]
    dy = rnd(B, N, D, seed=4)
    ref.backward(q(dy, dtype))
    out.backward(dy.to(DEV, dtype))
    assert rel_err(pd.grad, pr.grad) < TOL[dtype]
    assert rel_err(cd.grad, cr.grad) < 1e-5 and rel_err(ped.grad, per.grad) < 1e-5
    if masked:
        assert ped.grad[[2, 3, 9, 16]].abs().max() == 0 and ped.grad[N].abs().max() > 0


def test_tokens_gradient_is_deterministic(ops):
    B, N, D = 64, 65, 192
    dx = rnd(B, N, D, seed=5).to(DEV, torch.bfloat16)
    seq = torch.arange(N)
    seq[1::2] = N
    outs = []
    for _ in range(2):
        p = torch.zeros(B, N - 1, D, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        c = torch.zeros(1, 1, D, device=DEV, requires_grad=True)
        pe = torch.zeros(N + 1, D, device=DEV, requires_grad=True)
        ops.TokensFn.apply(p, c, pe, seq.to(DEV)).backward(dx)
        outs.append(pe.grad.clone())
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------ patch embedding
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("img,p,C,D,B", [(32, 4, 3, 96, 3), (64, 16, 3, 128, 2), (224, 16, 3, 64, 1), (24, 8, 1, 40, 2),
                                           (16, 2, 3, 24, 2), (32, 4, 3, 192, 64), (40, 4, 3, 64, 2), (20, 4, 3, 64, 2),
                                           (16, 1, 1, 16, 2), (32, 4, 1, 8, 1)])
def test_patch_embed(ops, dtype, img, p, C, D, B):
    # patches narrower than 8 pixels in bf16: groups of 8 / p patches read in place (image width a multiple of 8),
    # else (width 20) the gathered copy
    from visiontransformer_amd._lib import lib
    if dtype == torch.bfloat16 and p < 8:
        assert lib.vitmi_patch_wide_group(1, 1, B, C, img, img, p, D) == (8 // p if img % 8 == 0 else 0)
    x, w, b = rnd(B, C, img, img, seed=1), rnd(D, C, p, p, seed=2, scale=(C * p * p) ** -0.5), rnd(D, seed=3)
    xr = q(x, dtype).requires_grad_(True)
    wr, br = q(w, dtype).requires_grad_(True), b.clone().requires_grad_(True)
    ref = vit_oracle.patch_embed(xr, wr, br)
    xd = x.detach().to(DEV).requires_grad_(True)
    wd, bd = w.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    out = ops.PatchEmbedFn.apply(xd, wd, bd, dtype)
    assert out.shape == ref.shape and rel_err(out, ref) < TOL[dtype]
    dy = rnd(*ref.shape, seed=4)
    ref.backward(q(dy, dtype))
    out.backward(dy.to(DEV, dtype))
    assert rel_err(wd.grad, wr.grad) < (1e-4 if dtype == torch.float32 else 5e-3)
    assert rel_err(bd.grad, br.grad) < 1e-4
    assert rel_err(xd.grad, xr.grad) < TOL[dtype]


# ------------------------------------------------------------------------------------------ attention core
def attn_ref(qkv, H, DH):
    B, N, _ = qkv.shape
    I = H * DH
    q_, k_, v_ = (t.reshape(B, N, H, DH).permute(0, 2, 1, 3) for t in qkv.split(I, dim=-1))
    dots = (q_ @ k_.transpose(-1, -2)) * DH ** -0.5
    probs = torch.softmax(dots, -1)
    return (probs @ v_).permute(0, 2, 1, 3).reshape(B, N, I), probs, torch.logsumexp(dots, -1)


ATTN_CASES = [(2, 65, 12, 8), (2, 197, 3, 64), (1, 17, 2, 32), (3, 50, 4, 16), (1, 256, 2, 64), (2, 1, 2, 8),
              (1, 100, 1, 64), (2, 33, 2, 64)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,H,DH", ATTN_CASES)
def test_attention_forward_backward(ops, dtype, B, N, H, DH):
    qkv = rnd(B, N, 3 * H * DH, seed=1, scale=1.5)
    qr = q(qkv, dtype).requires_grad_(True)
    o_ref, p_ref, lse_ref = attn_ref(qr, H, DH)
    qd = qkv.to(DEV, dtype)
    o, lse, probs = ops.attn_fwd(qd, B, N, H, DH, True)
    assert rel_err(o, o_ref) < TOL[dtype]
    assert rel_err(probs, p_ref) < TOL[dtype]
    assert rel_err(lse, lse_ref) < 1e-4
    o2, _, none = ops.attn_fwd(qd, B, N, H, DH, False)
    assert none is None and torch.equal(o, o2)            # export does not change the numerics
    do = rnd(B, N, H * DH, seed=2)
    o_ref.backward(q(do, dtype))
    dqkv = ops.attn_bwd(qd, o, do.to(DEV, dtype), lse, B, N, H, DH)
    assert rel_err(dqkv, qr.grad) < TOL[dtype]


@pytest.mark.parametrize("B,N,H,DH", ATTN_CASES + [(1, 40, 2, 24)])
def test_attention_fp32_row_kernels(ops, B, N, H, DH):
    """The row-by-row fp32 kernels (what an fp32 shape outside the matrix-instruction kernels' range takes, e.g. a
    head dim of 24) against the fp32 expression, and the two families against each other where both apply."""
    from visiontransformer_amd._lib import lib
    dtype = torch.float32
    qkv = rnd(B, N, 3 * H * DH, seed=11, scale=1.5)
    do = rnd(B, N, H * DH, seed=12)
    qr = q(qkv, dtype).requires_grad_(True)
    o_ref, p_ref, lse_ref = attn_ref(qr, H, DH)
    o_ref.backward(q(do, dtype))
    qd, dod = qkv.to(DEV, dtype), do.to(DEV, dtype)
    res = {}
    try:
        for mode in (0, 1):
            lib.vitmi_debug_set_attn_f32_mfma(mode)
            o, lse, probs = ops.attn_fwd(qd, B, N, H, DH, True)
            dqkv = ops.attn_bwd(qd, o, dod, lse, B, N, H, DH)
            assert rel_err(o, o_ref) < 1e-4 and rel_err(probs, p_ref) < 1e-4 and rel_err(lse, lse_ref) < 1e-4
            assert rel_err(dqkv, qr.grad) < 1e-4
            res[mode] = (o, lse, probs, dqkv)
    finally:
        lib.vitmi_debug_set_attn_f32_mfma(1)
    for a, b in zip(res[0], res[1]):
        assert rel_err(a, b.cpu()) < 2e-5


def test_attention_spiky_scores(ops):
    # one key dominates a row by a wide margin, another row is all-equal: exercises max-subtraction
    B, N, H, DH = 1, 70, 1, 64
    qkv = rnd(B, N, 3 * DH, seed=3, scale=0.1)
    qkv[0, 5, :DH] = 6.0
    qkv[0, 40, DH:2 * DH] = 6.0
    qkv[0, 7, :DH] = 0.0
    for dtype in (torch.float32, torch.bfloat16):
        o_ref, p_ref, _ = attn_ref(q(qkv, dtype), H, DH)
        o, lse, probs = ops.attn_fwd(qkv.to(DEV, dtype), B, N, H, DH, True)
        assert torch.isfinite(o.float()).all() and rel_err(o, o_ref) < TOL[dtype]
        assert rel_err(probs, p_ref) < TOL[dtype]
        assert abs(probs[0, 0, 7].float().sum().item() - 1.0) < 1e-2


@pytest.mark.parametrize("B,N,H,DH", [(128, 65, 12, 8), (96, 33, 18, 16), (260, 17, 6, 8)])
def test_attention_head_groups_bf16(ops, B, N, H, DH):
    # short sequences of narrow heads with >= 256 groups of six heads: a workgroup takes six heads of an image
    # (attn_*_kernel<.., HPW = 6>); against the fp32 expression and, bit for bit, against one pair per workgroup
    from visiontransformer_amd._lib import lib
    dtype = torch.bfloat16
    qkv = rnd(B, N, 3 * H * DH, seed=5, scale=1.5)
    do = rnd(B, N, H * DH, seed=6)
    qr = q(qkv, dtype).requires_grad_(True)
    o_ref, p_ref, lse_ref = attn_ref(qr, H, DH)
    o_ref.backward(q(do, dtype))
    qd = qkv.to(DEV, dtype)
    got = {}
    try:
        for on in (1, 0):
            lib.vitmi_debug_set_attn_head_groups(on)
            o, lse, probs = ops.attn_fwd(qd, B, N, H, DH, True)
            o2, _, _ = ops.attn_fwd(qd, B, N, H, DH, False)
            dqkv = ops.attn_bwd(qd, o, do.to(DEV, dtype), lse, B, N, H, DH)
            got[on] = (o, lse, probs, o2, dqkv)
    finally:
        lib.vitmi_debug_set_attn_head_groups(1)
    o, lse, probs, o2, dqkv = got[1]
    assert rel_err(o, o_ref) < TOL[dtype] and rel_err(probs, p_ref) < TOL[dtype] and rel_err(lse, lse_ref) < 1e-4
    assert torch.equal(o, o2) and rel_err(dqkv, qr.grad) < TOL[dtype]
    assert all(torch.equal(a, b) for a, b in zip(got[1], got[0]))


@pytest.mark.parametrize("B,N,H,DH", [(40, 197, 12, 64), (3, 129, 2, 64), (5, 160, 3, 64), (2, 192, 2, 64),
                                      (90, 193, 3, 64), (2, 208, 5, 64), (1, 200, 1, 64)])
def test_attention_backward_one_kernel_bf16(ops, B, N, H, DH):
    # heads of 64 with 129..208 tokens: the backward runs as ONE kernel (attn_bwd_fused_bf16_kernel; with 193..208
    # tokens attn_bwd_pipe_bf16_kernel, which fetches the next pair under the current one; more pairs than compute
    # units make a workgroup walk several).  Each variant and the dQ + dK/dV pair against the fp32 expression.
    from visiontransformer_amd._lib import lib
    dtype = torch.bfloat16
    qkv = rnd(B, N, 3 * H * DH, seed=7, scale=1.5)
    do = rnd(B, N, H * DH, seed=8)
    qr = q(qkv, dtype).requires_grad_(True)
    o_ref, _, _ = attn_ref(qr, H, DH)
    o_ref.backward(q(do, dtype))
    qd, dod = qkv.to(DEV, dtype), do.to(DEV, dtype)
    o, lse, _ = ops.attn_fwd(qd, B, N, H, DH, False)
    try:
        for mode in (0, 1, 2):
            lib.vitmi_debug_set_attn_bwd_fused(mode)
            dqkv = ops.attn_bwd(qd, o, dod, lse, B, N, H, DH).clone()
            again = ops.attn_bwd(qd, o, dod, lse, B, N, H, DH)
            assert torch.equal(dqkv, again), f"mode {mode}: two calls differ"
            assert rel_err(dqkv, qr.grad) < TOL[dtype], f"mode {mode}"
    finally:
        lib.vitmi_debug_set_attn_bwd_fused(2)


@pytest.mark.parametrize("B,N,H,DH", [(40, 197, 12, 64), (3, 129, 2, 64), (6, 100, 3, 64), (2, 300, 2, 64), (70, 208, 4, 64)])
def test_attention_head_major_planes_bf16(ops, B, N, H, DH):
    # qkv / dqkv as head-major planes [3][H][B*N][64] (what the whole-block calls use for heads of 64): the same
    # kernels on another addressing, so every result equals the row-layout one bit for bit -- forward (with and without
    # export), every backward variant, planes in, planes out, and both
    from visiontransformer_amd._lib import check, lib
    dtype = torch.bfloat16
    T, I = B * N, H * DH

    def to_planes(t):        # [B, N, 3*I] -> [3, H, T, DH] flattened into the same number of elements
        return t.view(T, 3, H, DH).permute(1, 2, 0, 3).contiguous().view(B, N, 3 * I)

    def from_planes(t):
        return t.view(3, H, T, DH).permute(2, 0, 1, 3).contiguous().view(B, N, 3 * I)

    qd = rnd(B, N, 3 * I, seed=11, scale=1.5).to(DEV, dtype)
    dod = rnd(B, N, I, seed=12).to(DEV, dtype)
    qp = to_planes(qd)
    assert torch.equal(from_planes(qp), qd)
    export = N <= 256
    o, lse, probs = ops.attn_fwd(qd, B, N, H, DH, export)
    o2, lse2, probs2 = ops.attn_fwd(qp, B, N, H, DH, export, layout=1)
    assert torch.equal(o, o2) and torch.equal(lse, lse2) and (not export or torch.equal(probs, probs2))
    p = ops._p
    try:
        for mode in (0, 1, 2):
            lib.vitmi_debug_set_attn_bwd_fused(mode)
            want = ops.attn_bwd(qd, o, dod, lse, B, N, H, DH)
            for layout in (1, 2, 3):
                src = qp if layout & 1 else qd
                got = torch.full_like(qd, float("nan"))
                ws = ops._ws(lib.vitmi_attn_bwd_workspace_bytes(1, B, N, H, DH), DEV)
                check(lib.vitmi_attn_bwd(p(src), p(o), p(dod), p(lse), None, p(got), p(ws), 1, B, N, H, DH, layout,
                                         ops._stream()), "vitmi_attn_bwd")
                got = from_planes(got) if layout & 2 else got
                assert torch.equal(got, want), (mode, layout)
    finally:
        lib.vitmi_debug_set_attn_bwd_fused(2)


def test_attention_planes_need_bf16_heads_of_64(ops):
    from visiontransformer_amd._lib import VitmiError
    with pytest.raises(VitmiError, match="head-major"):
        ops.attn_fwd(torch.zeros(2, 65, 3 * 96, device=DEV, dtype=torch.bfloat16), 2, 65, 12, 8, False, layout=1)
    with pytest.raises(VitmiError, match="head-major"):
        ops.attn_fwd(torch.zeros(2, 65, 3 * 64, device=DEV), 2, 65, 1, 64, False, layout=1)


def test_attention_backward_one_kernel_with_reserved_compute_units(ops):
    # vitmi_set_reserved_cus leaves compute units to a collective: the one-workgroup-per-CU backward launches that many
    # fewer workgroups and each walks more pairs; same gradients bit for bit
    from visiontransformer_amd._lib import check, lib
    B, N, H, DH = 64, 197, 12, 64
    qd = rnd(B, N, 3 * H * DH, seed=9, scale=1.5).to(DEV, torch.bfloat16)
    dod = rnd(B, N, H * DH, seed=10).to(DEV, torch.bfloat16)
    o, lse, _ = ops.attn_fwd(qd, B, N, H, DH, False)
    full = ops.attn_bwd(qd, o, dod, lse, B, N, H, DH).clone()
    try:
        for k in (8, 64):
            check(lib.vitmi_set_reserved_cus(k), "vitmi_set_reserved_cus")
            assert torch.equal(full, ops.attn_bwd(qd, o, dod, lse, B, N, H, DH)), k
    finally:
        check(lib.vitmi_set_reserved_cus(0), "vitmi_set_reserved_cus")


def test_attention_rejects_long_sequences(ops):
    from visiontransformer_amd._lib import VitmiError
    with pytest.raises(VitmiError, match="not supported"):
        ops.attn_fwd(torch.zeros(1, 609, 3 * 64, device=DEV, dtype=torch.bfloat16), 1, 609, 1, 64, False)
    with pytest.raises(VitmiError, match="not supported"):        # beyond 256 tokens only bf16 heads of 64
        ops.attn_fwd(torch.zeros(1, 300, 3 * 32, device=DEV, dtype=torch.bfloat16), 1, 300, 1, 32, False)
    with pytest.raises(VitmiError, match="not supported"):
        ops.attn_fwd(torch.zeros(1, 300, 3 * 64, device=DEV), 1, 300, 1, 64, False)


# ------------------------------------------------------------------------------------------ misc
def test_cast_add_meanpool(ops):
    x = rnd(1001, seed=1).to(DEV)
    xb = ops.cast(x, torch.bfloat16)
    assert torch.equal(xb.cpu(), x.cpu().to(torch.bfloat16))       # RNE, bit-exact vs torch's CPU cast
    assert torch.equal(ops.cast(xb, torch.float32).cpu(), xb.cpu().float())
    a, b = rnd(777, seed=2).to(DEV), rnd(777, seed=3).to(DEV)
    assert torch.equal(ops.add(a, b).cpu(), a.cpu() + b.cpu())
    t = rnd(3, 17, 40, seed=4).to(DEV).requires_grad_(True)
    m = ops.MeanPoolFn.apply(t)
    assert rel_err(m, t.detach().cpu().mean(1)) < 1e-6
    m.sum().backward()
    assert rel_err(t.grad, torch.full((3, 17, 40), 1 / 17)) < 1e-6


# ------------------------------------------------------------------------------------------ aligned fast paths
# the last three have >= 256 output tiles and K % 64 == 0: the persistent phased kernel (ragged M and N
# edges, one / two / three K-tiles, blocks that walk more than one tile)
@pytest.mark.parametrize("M,N,K", [(394, 768, 768), (1000, 2304, 768), (520, 768, 3072), (129, 136, 72), (5, 8, 8),
                                   (256, 256, 4096), (16500, 1032, 192), (20000, 1024, 64), (20000, 1024, 128)])
def test_fast_nt_gemm_all_epilogues(ops, M, N, K):
    dtype = torch.bfloat16
    x, w, b, r = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23), rnd(M, N, seed=24)
    xd, wd, bd, rd = x.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV), r.to(DEV, dtype)
    pre = q(x, dtype) @ q(w, dtype).t() + b
    y, _ = ops.linear_fwd(xd, wd, bd, res=rd)
    assert rel_err(y, pre + q(r, dtype)) < 1e-2
    h, u = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_GELU, want_aux=True)
    assert rel_err(u, pre) < 1e-2 and rel_err(h, vit_oracle.gelu_erf(pre)) < 1e-2
    y32, _ = ops.linear_fwd(xd, wd, None, out_dtype=torch.float32)
    assert rel_err(y32, pre - b) < 2e-3
    # input gradient through the transposed weight copy, with and without the dGELU epilogue
    wp = torch.nn.Parameter(w.to(DEV))
    wt = ops.compute_copy_t(wp, dtype)
    assert torch.equal(wt.cpu(), w.to(dtype).t().contiguous())
    dy = rnd(M, N, seed=25)
    dx = ops.linear_bwd_x(dy.to(DEV, dtype), wd, w_t=wt)
    assert rel_err(dx, q(dy, dtype) @ q(w, dtype)) < 1e-2
    aux = rnd(M, K, seed=26).to(DEV, dtype)
    du = ops.linear_bwd_x(dy.to(DEV, dtype), wd, epi=ops.EPI_DGELU, aux=aux, w_t=wt)
    a = aux.float().cpu().requires_grad_(True)
    vit_oracle.gelu_erf(a).backward(q(dy, dtype) @ q(w, dtype))
    assert rel_err(du, a.grad) < 1e-2
    # the same product with the column sums of its result (bias gradient of the layer below) from one call
    du2, cs = ops.linear_bwd_x(dy.to(DEV, dtype), wd, epi=ops.EPI_DGELU, aux=aux, w_t=wt, want_colsum=True)
    assert torch.equal(du2, du)
    assert rel_err(cs, a.grad.sum(0)) < 1e-2
    assert rel_err(cs, du.float().sum(0)) < 2e-3          # sums of the unrounded values vs sums of bf16
    _, cs2 = ops.linear_bwd_x(dy.to(DEV, dtype), wd, epi=ops.EPI_DGELU, aux=aux, w_t=wt, want_colsum=True)
    assert torch.equal(cs, cs2)                            # fixed summation order
    # derivative stored by the forward epilogue (GELU_D), multiplied in by the backward one (MULAUX)
    h2, gp = ops.linear_fwd(xd, wd, bd, epi=ops.EPI_GELU_D, want_aux=True)
    pr = pre.clone().requires_grad_(True)
    vit_oracle.gelu_erf(pr).sum().backward()
    # <GELU> evaluates the erf form, <GELU_D> the fitted sigmoid form (3.8e-5 apart before rounding): equal up to
    # bf16 units in the last place (3.9e-3 of the largest value each)
    assert rel_err(h2, h) < 8e-3 and rel_err(deriv(gp), pr.grad) < 1e-2
    auxh = aux.float().half().view(torch.bfloat16)                 # the pair keeps its aux as IEEE half bits
    dm = ops.linear_bwd_x(dy.to(DEV, dtype), wd, epi=ops.EPI_MULAUX, aux=auxh, w_t=wt)
    assert rel_err(dm, (q(dy, dtype) @ q(w, dtype)) * deriv(auxh).cpu()) < 1e-2
    dm2, csm = ops.linear_bwd_x(dy.to(DEV, dtype), wd, epi=ops.EPI_MULAUX, aux=auxh, w_t=wt, want_colsum=True)
    assert torch.equal(dm2, dm) and rel_err(csm, dm.float().sum(0)) < 2e-3


# T % 64 == 0 with P, Q >= 512 takes the phased kernel (ragged P and Q edges, one and two token tiles)
@pytest.mark.parametrize("T,P,Q", [(6304, 768, 768), (1576, 2304, 768), (12608, 768, 3072), (197, 96, 192), (64, 8, 8),
                                   (1000, 136, 520), (640, 520, 1032), (64, 768, 768), (128, 512, 520)])
def test_fast_weight_gradient_gemm(ops, T, P, Q):
    dtype = torch.bfloat16
    dy, x = rnd(T, P, seed=31), rnd(T, Q, seed=32)
    dw = ops.linear_bwd_w(dy.to(DEV, dtype), x.to(DEV, dtype))
    ref = q(dy, dtype).double().t() @ q(x, dtype).double()
    assert rel_err(dw, ref) < 1e-4                      # fp32 accumulation of exact bf16 products
    dw2 = ops.linear_bwd_w(dy.to(DEV, dtype), x.to(DEV, dtype))
    assert torch.equal(dw, dw2)                         # slab order is fixed: bitwise reproducible


# ------------------------------------------------------------------------------------------ optimizer
def test_fused_adam_matches_torch_adam(ops):
    """One-launch multi-tensor Adam against torch.optim.Adam (trainer.py:204-205 settings) on tensors
    that are smaller than, equal to and larger than a chunk, with sizes not divisible by 4."""
    from visiontransformer_amd.optim import CHUNK, FusedAdam
    shapes = [(7,), (1, 1, 3), (CHUNK,), (CHUNK + 5,), (3, CHUNK // 2 + 1), (768, 96)]
    ref_p = [torch.nn.Parameter(rnd(*s, seed=40 + i)) for i, s in enumerate(shapes)]
    our_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    kw = dict(lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)
    ref_opt, our_opt = torch.optim.Adam(ref_p, **kw), FusedAdam(our_p, **kw)
    # a bf16 compute copy that the kernel must keep in step with the master weights
    w16 = ops.compute_copy(our_p[-1], torch.bfloat16)
    for step in range(3):
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            g = rnd(*a.shape, seed=100 + 10 * step + i)
            a.grad, b.grad = g, g.to(DEV)
        ref_opt.step()
        our_opt.step()
    for a, b in zip(ref_p, our_p):
        assert rel_err(b, a.detach()) < 1e-6
    assert ops.compute_copy(our_p[-1], torch.bfloat16) is w16
    assert torch.equal(w16.cpu(), our_p[-1].detach().cpu().to(torch.bfloat16))


@pytest.mark.parametrize("mdt", [torch.int64, torch.int32])
def test_seq_ord_kernel_bit_exact_against_reference_vectors(mdt):
    """vitmi_seq_ord against the seq_ord arrays captured from the reference's index arithmetic (masks.npz)."""
    import os
    from conftest import GOLDEN
    from visiontransformer_amd import ops
    z = np.load(os.path.join(GOLDEN, "masks.npz"), allow_pickle=False)
    for i in range(int(z["count"])):
        um = torch.from_numpy(z[f"{i}/unk_mask"]).to(mdt)
        n = um.shape[1]
        got = ops.seq_ord(um.to(DEV), n, n, DEV)
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy()[None, :], z[f"{i}/seq_ord"])
        neg = ops.seq_ord(um, n, -1, DEV).cpu().numpy()                  # host mask, "no row" marker
        m = z[f"{i}/unk_mask"].reshape(-1).astype(np.int64)
        assert np.array_equal(neg, np.arange(n) * (1 - m) - m)
    with pytest.raises(ValueError):
        ops.seq_ord(torch.zeros(1, 5, dtype=torch.long, device=DEV), 6, 6, DEV)


def test_upload_ring_reuses_its_slots_safely():
    """ops.upload: more uploads than staging slots, every array arrives intact (a slot is rewritten only after the
    copy that read it has finished), for the three item types."""
    from visiontransformer_amd import ops
    rng = np.random.default_rng(0)
    sent, got = [], []
    for i in range(3 * ops._UPLOAD_SLOTS + 1):
        dt = (np.int32, np.int64, np.float32)[i % 3]
        a = (rng.standard_normal((3, 1 + 37 * i)) * 1000).astype(dt)
        sent.append(a)
        got.append(ops.upload(a, DEV))
    torch.cuda.synchronize()
    for a, g in zip(sent, got):
        assert tuple(g.shape) == a.shape and np.array_equal(g.cpu().numpy(), a)
    big = rng.integers(0, 1 << 30, size=200000).astype(np.int64)          # larger than a slot's initial size
    assert np.array_equal(ops.upload(big, DEV).cpu().numpy(), big)


@pytest.mark.parametrize("tile_form", [True, False])
def test_fused_adam_refreshes_transposed_copies(tile_form, monkeypatch):
    """After FusedAdam.step the cached W^T bf16 copies and the straight copies match the updated fp32 parameters, for
    ragged and aligned shapes: written by the tile form of the update itself (vitmi_adam_step_matrices: columns a
    multiple of 4, rows a multiple of 4 or not, whole and partial 64 x 64 tiles) or, for the other matrices and with
    the tile form switched off, by the transpose pass (vitmi_transpose_cast_multi).  Both forms leave the same bits in
    the parameters and the optimizer state."""
    from visiontransformer_amd import ops, optim
    from visiontransformer_amd.optim import FusedAdam
    shapes = [(96, 64), (33, 70), (256, 257), (5, 3), (128, 100), (130, 68), (64, 192), (3, 8)]

    def run(tile):
        monkeypatch.setattr(optim, "TILE_ADAM", tile)
        ops.clear_cast_cache()
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(r, c, device=DEV)) for r, c in shapes]
        ps.append(torch.nn.Parameter(torch.randn(17, device=DEV)))
        opt = FusedAdam(ps, lr=1e-2, weight_decay=1e-2)
        for step in range(3):
            straight = [ops.compute_copy(p, torch.bfloat16) for p in ps]
            transposed = [ops.compute_copy_t(p, torch.bfloat16) for p in ps[:len(shapes)]]
            for p in ps:
                p.grad = torch.randn_like(p)
            before = [p.detach().clone() for p in ps]
            opt.step()
            for p, b, s in zip(ps, before, straight):
                assert not torch.equal(p, b)
                assert ops.compute_copy(p, torch.bfloat16) is s and torch.equal(s, p.detach().to(torch.bfloat16))
            for p, t in zip(ps, transposed):
                assert ops.compute_copy_t(p, torch.bfloat16) is t                 # refreshed in place, not rebuilt
                assert torch.equal(t, p.detach().t().contiguous().to(torch.bfloat16))
        return [p.detach().clone() for p in ps], [tuple(x.clone() for x in opt.state[p]) for p in ps]

    params, state = run(tile_form)
    if tile_form:
        ref_params, ref_state = run(False)
        for a, b in zip(params, ref_params):
            assert torch.equal(a, b)
        for (m1, v1), (m2, v2) in zip(state, ref_state):
            assert torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.parametrize("N", [257, 384, 400, 577, 608])
def test_attention_long_sequences_bf16(N):
    """Sequences of 257..608 tokens (heads of 64, bf16): the whole head still sits in LDS, chunk counts 12 / 16 / 19
    with the chunks beyond N masked.  Against the fp32 expression of mymodel/ViT.py:98-102 and autograd over it,
    evaluated on the CPU from the bf16-rounded inputs."""
    from visiontransformer_amd import ops
    B, H, DH = 2, 3, 64
    gen = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3 * H * DH, generator=gen).to(DEV).bfloat16()
    do = torch.randn(B, N, H * DH, generator=gen).to(DEV).bfloat16()
    ref_in = qkv.float().cpu().requires_grad_(True)
    q, k, v = (t.reshape(B, N, H, DH).permute(0, 2, 1, 3) for t in ref_in.split(H * DH, dim=-1))
    p = torch.softmax((q @ k.transpose(-1, -2)) * DH ** -0.5, dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, N, H * DH)
    ref.backward(do.float().cpu())
    o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
    assert rel_err(o, ref.detach()) < 2e-2
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)
    assert rel_err(dqkv, ref_in.grad) < 2e-2
    with pytest.raises(RuntimeError, match="export"):
        ops.attn_fwd(qkv, B, N, H, DH, True)


def test_fused_adam_groups_skipped_steps_and_type_checks():
    """param_groups with their own learning rates (a scheduler writes group["lr"]), a parameter that skips a step
    keeps torch.optim.Adam's per-parameter bias correction, and anything but contiguous float32 is refused instead
    of being overwritten past its storage."""
    from visiontransformer_amd.optim import FusedAdam
    ref_p = [torch.nn.Parameter(rnd(33, 7, seed=60 + i)) for i in range(3)]
    our_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]

    def groups(ps):
        return [{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 1e-3, "weight_decay": 0.1}]

    ref_opt = torch.optim.Adam(groups(ref_p), betas=(0.9, 0.99))
    our_opt = FusedAdam(groups(our_p), betas=(0.9, 0.99))
    assert [g["lr"] for g in our_opt.param_groups] == [1e-2, 1e-3]
    for step in range(4):
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            if i == 1 and step == 1:            # this parameter takes no gradient at step 1
                a.grad = b.grad = None
                continue
            g = rnd(33, 7, seed=200 + 10 * step + i)
            a.grad, b.grad = g, g.to(DEV)
        if step == 2:                           # what a scheduler does between epochs
            ref_opt.param_groups[0]["lr"] = our_opt.param_groups[0]["lr"] = 5e-3
        ref_opt.step()
        our_opt.step()
    for a, b in zip(ref_p, our_p):
        assert rel_err(b, a.detach()) < 1e-6
    half = torch.nn.Parameter(torch.zeros(8, 8, device=DEV, dtype=torch.bfloat16))
    half.grad = torch.ones_like(half)
    with pytest.raises(TypeError, match="contiguous float32"):
        FusedAdam([half]).step()
    strided = torch.nn.Parameter(torch.zeros(8, 8, device=DEV).t())
    strided.grad = torch.ones(8, 8, device=DEV)
    with pytest.raises(TypeError, match="contiguous float32"):
        FusedAdam([strided]).step()
