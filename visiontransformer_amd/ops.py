"""torch.autograd glue over the C ABI of libvitmi_hip.so.

Every function here moves raw device pointers, shapes and the current HIP stream across the
boundary; tensors are allocated by PyTorch's caching allocator.  There is no CPU or eager
fallback: CPU tensors raise.
"""
import os
import weakref

import torch

from . import _lib
from ._lib import BF16, EPI_DGELU, EPI_GELU, EPI_GELU_D, EPI_MULAUX, EPI_NONE, F32, check, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError(f"visiontransformer_amd computes in float32 or bfloat16, not {t.dtype}") from None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "visiontransformer_amd runs only on ROCm device tensors (HIP kernels for gfx950); "
                "got a CPU tensor and there is no CPU fallback")


def _p(t):
    return None if t is None else t.data_ptr()


# the raw handle of the current stream of the current device; the private getters skip the Stream object that
# torch.cuda.current_stream() builds on every call (several microseconds, once per kernel launch)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------------------------
# small host -> device tables
# ---------------------------------------------------------------------------------------------
_UPLOAD_SLOTS = 8
_upload_ring = {}


def upload(arr, device):
    """A numpy array of 4- or 8-byte items as a device tensor, copied in stream order from a pinned staging slot
    so the host does not wait for the stream (a copy from pageable memory would).  A slot is reused only after
    the copy that last read it has finished."""
    import numpy as np
    arr = np.ascontiguousarray(arr)
    tdt = {"int32": torch.int32, "int64": torch.int64, "float32": torch.float32}[arr.dtype.name]
    device = torch.device(device)
    ring = _upload_ring.get(device)
    if ring is None:
        ring = _upload_ring[device] = {"host": [None] * _UPLOAD_SLOTS, "ev": [None] * _UPLOAD_SLOTS, "i": 0}
    i = ring["i"]
    ring["i"] = (i + 1) % _UPLOAD_SLOTS
    if ring["ev"][i] is not None:
        ring["ev"][i].synchronize()
    n = arr.nbytes
    if ring["host"][i] is None or ring["host"][i].numel() < n:
        ring["host"][i] = torch.empty(max(n, 1 << 14), dtype=torch.uint8).pin_memory()
    stage = ring["host"][i][:n]
    stage.numpy()[:] = arr.reshape(-1).view(np.uint8)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        out.copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    ring["ev"][i] = ev
    return out.view(tdt).view(arr.shape)


# ---------------------------------------------------------------------------------------------
# compute-dtype copies of fp32 parameters, refreshed when the parameter's version changes
# ---------------------------------------------------------------------------------------------
_cast_cache = {}


def cast(t, dtype):
    """t converted to `dtype` by the HIP cast kernel (t itself when it already matches)."""
    if t.dtype == dtype:
        return t
    _need_gpu(t)
    src = t.detach().contiguous()
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib.vitmi_cast(_p(src), _dt(src), _p(out), _dt(dtype), src.numel(), _stream()), "vitmi_cast")
    return out


def _cache_lookup(key, param):
    """The cached copy for `key` if it was made from this very tensor object at its current version
    and storage (id() values are recycled once a tensor dies, hence the weak reference)."""
    hit = _cast_cache.get(key)
    if hit is None:
        return None
    ver, ptr, copy, ref = hit
    if ref() is param and ver == param._version and ptr == param.data_ptr() and copy.device == param.device:
        return copy
    return None


def _cache_store(key, param, copy):
    def _drop(_ref, key=key):
        ent = _cast_cache.get(key)
        if ent is not None and ent[3] is _ref:
            del _cast_cache[key]
    _cast_cache[key] = (param._version, param.data_ptr(), copy, weakref.ref(param, _drop))


def compute_copy(param, dtype):
    """A `dtype` copy of parameter `param`, cached until the parameter is next modified in place."""
    if param.dtype == dtype:
        return param.detach()
    key = (id(param), dtype)
    hit = _cache_lookup(key, param)
    if hit is not None:
        return hit
    c = cast(param, dtype)
    _cache_store(key, param, c)
    return c


def compute_copy_t(param, dtype):
    """Transposed compute copy of a 2-D weight: [K, N] for a parameter stored [N, K].  bf16 only
    (the fp32 kernels read the weight through strides instead)."""
    key = (id(param), dtype, "T")
    hit = _cache_lookup(key, param)
    if hit is not None:
        return hit
    _need_gpu(param)
    src = param.detach().contiguous()
    n, k = src.shape
    out = torch.empty((k, n), dtype=dtype, device=src.device)
    check(lib.vitmi_transpose_cast(_p(src), _dt(src), _p(out), n, k, _stream()), "vitmi_transpose_cast")
    _cache_store(key, param, out)
    return out


# LayerNorm folded into the weight of the Linear behind it, for the inference entry points (vitmi_gemm_ln):
# (Wg, s, c) = (bf16(W * gamma), row sums of Wg in fp32, W beta + b in fp32), cached until one of the four
# parameters changes.  Weight preparation, like the compute copies above; a few launches per weight version.
_fold_cache = {}
_infer_fusion = os.environ.get("VITMI_INFER_FUSION", "1") != "0"


def set_gelu(mode):
    """How the bf16 MLP block evaluates nn.GELU() and its derivative (process-wide; fp32 mode always uses erff):
    "fast" (default) = gelu and gelu' from one fitted sigmoid form in the forward epilogue (errors 3.8e-5 / 9.3e-5
    against the erf forms, below bf16 rounding), "erf" = the erf form (1.5e-7) with the derivative evaluated from the
    stored pre-activation in the backward pass.  Returns the previous mode.  Do not switch between a forward pass and
    its backward pass."""
    if mode not in ("fast", "erf"):
        raise ValueError('set_gelu: mode must be "fast" or "erf"')
    prev = "erf" if lib.vitmi_get_gelu_mode() else "fast"
    check(lib.vitmi_set_gelu_mode(int(mode == "erf")), "vitmi_set_gelu_mode")
    return prev


def set_inference_fusion(on):
    """Whether forward passes that build no autograd graph take the LayerNorm-folded inference entry points
    (default on; VITMI_INFER_FUSION=0 turns it off at import).  Returns the previous setting."""
    global _infer_fusion
    prev, _infer_fusion = _infer_fusion, bool(on)
    return prev


_param_epoch = 0


def param_epoch():
    """Counts the events that change parameters behind autograd's back (raw-pointer writes of the fused optimizer
    kernel, which move no version counter): consumers that captured derived operands compare it."""
    return _param_epoch


def drop_folded():
    """Forget every folded operand (call after parameters were changed without a version bump)."""
    global _param_epoch
    _param_epoch += 1
    _fold_cache.clear()


def folded_linear(w, ln_w, ln_b, b):
    parts = [t for t in (w, ln_w, ln_b, b) if t is not None]
    key = tuple(id(t) for t in (w, ln_w, ln_b, b))
    stamp = tuple((t._version, t.data_ptr()) for t in parts)
    hit = _fold_cache.get(key)
    if hit is not None and hit[0] == stamp and all(r() is t for r, t in zip(hit[1], parts)):
        return hit[2]
    _need_gpu(w, ln_w, ln_b, b)
    n, k = w.shape
    w32, g32, b32 = (t.detach().float().contiguous() for t in (w, ln_w, ln_b))
    bias32 = b.detach().float().contiguous() if b is not None else None
    wg = torch.empty((n, k), dtype=torch.bfloat16, device=w.device)
    s = torch.empty(n, dtype=torch.float32, device=w.device)
    c = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.vitmi_fold_ln(_p(w32), _p(g32), _p(b32), _p(bias32), _p(wg), _p(s), _p(c), n, k, _stream()),
          "vitmi_fold_ln")

    def _drop(_ref, key=key):
        _fold_cache.pop(key, None)
    _fold_cache[key] = (stamp, [weakref.ref(t, _drop) for t in parts], (wg, s, c))
    return wg, s, c


def pinned_compute_copies(params):
    """(key, parameter, copy) of every cached compute copy of the given parameters: the buffers a captured graph
    reads the weights from."""
    wanted = {id(p): p for p in params}
    out = []
    for key, (_, _, copy, ref) in list(_cast_cache.items()):
        p = wanted.get(key[0])
        if p is not None and ref() is p:
            out.append((key, p, copy))
    return out


def refresh_pinned_copies(pinned):
    """Bring pinned compute copies up to date IN PLACE.  A captured graph keeps reading the buffers it was captured
    with; when a parameter was modified since (an optimizer that goes through autograd-visible in-place ops, a
    load_state_dict) or the cache was cleared, the cache would hand eager code a NEW buffer and the graph would go
    on computing with the old weights.  This rewrites the old buffers and re-registers them as the current copies."""
    for key, p, copy in pinned:
        ent = _cast_cache.get(key)
        if (ent is not None and ent[2] is copy and ent[3]() is p and ent[0] == p._version
                and ent[1] == p.data_ptr()):
            continue
        src = p.detach().contiguous()
        if len(key) == 3:       # transposed copy
            n, k = src.shape
            check(lib.vitmi_transpose_cast(_p(src), _dt(src), _p(copy), n, k, _stream()), "vitmi_transpose_cast")
        else:
            check(lib.vitmi_cast(_p(src), _dt(src), _p(copy), _dt(copy.dtype), src.numel(), _stream()), "vitmi_cast")
        _cache_store(key, p, copy)


def patch_wide_weight(weight, bias, g):
    """(block weight [g*D, C*p*8] bf16, block bias [g*D] fp32 or None) of vitmi_patch_wide_weight for a Conv2d weight
    [D, C, p, p] with p < 8.  Built on every call (one launch over 74 KB at the 32-pixel geometry): a cached copy would
    go stale inside a captured graph, whose replays see the parameters the optimizer updates in place."""
    D, C, p, _ = weight.shape
    w32 = weight.detach().float().contiguous()
    b32 = bias.detach().float().contiguous() if bias is not None else None
    w_wide = torch.empty((g * D, C * p * 8), dtype=torch.bfloat16, device=weight.device)
    b_wide = torch.empty(g * D, dtype=torch.float32, device=weight.device) if bias is not None else None
    check(lib.vitmi_patch_wide_weight(_p(w32), _p(b32), _p(w_wide), _p(b_wide), D, C, p, _stream()),
          "vitmi_patch_wide_weight")
    return w_wide, b_wide


def clear_cast_cache():
    """Drop every cached compute copy (call after parameters were changed behind autograd's back,
    e.g. by the fused optimizer kernel)."""
    global _param_epoch
    _param_epoch += 1
    _cast_cache.clear()
    _fold_cache.clear()


# ---------------------------------------------------------------------------------------------
# thin wrappers (no autograd)
# ---------------------------------------------------------------------------------------------
def gemm_raw(dt, M, N, K, A, sam, sak, B, sbn, sbk, C, ldc, c_dt, epi=EPI_NONE, bias=None, res=None, ldr=0,
             aux=None, ldaux=0):
    check(lib.vitmi_gemm(dt, M, N, K, _p(A), sam, sak, _p(B), sbn, sbk, _p(C), ldc, c_dt, epi, _p(bias), _p(res), ldr,
                         _p(aux), ldaux, _stream()), "vitmi_gemm")


def linear_fwd(x2, w, bias=None, res=None, epi=EPI_NONE, want_aux=False, out_dtype=None):
    """y[T,N] = x2[T,K] @ w[N,K]^T (+bias) (gelu) (+res).  Returns (y, aux) ; aux = pre-activation."""
    T, K = x2.shape
    N = w.shape[0]
    out_dtype = out_dtype or x2.dtype
    y = torch.empty((T, N), dtype=out_dtype, device=x2.device)
    aux = torch.empty((T, N), dtype=x2.dtype, device=x2.device) if want_aux else None
    gemm_raw(_dt(x2), T, N, K, x2, x2.stride(0), 1, w, w.stride(0), 1, y, N, _dt(y), epi, bias, res,
             res.stride(0) if res is not None else 0, aux, N if want_aux else 0)
    return y, aux


def linear_bwd_x(dy2, w, epi=EPI_NONE, aux=None, w_t=None, want_colsum=False):
    """dx[T,K] = dy2[T,N] @ w[N,K]   (optionally times gelu'(aux)).  With w_t = w^T stored [K,N]
    the product runs in the k-contiguous form.  want_colsum also returns the fp32 column sums of
    dx (the bias gradient of the layer below) from the same launch."""
    T, N = dy2.shape
    K = w.shape[1] if w_t is None else w_t.shape[0]
    dx = torch.empty((T, K), dtype=dy2.dtype, device=dy2.device)
    if want_colsum:
        _need_gpu(dy2)
        cs = torch.empty((K,), dtype=torch.float32, device=dy2.device)
        ws = _ws(lib.vitmi_gemm_colsum_workspace_bytes(T, K), dy2.device)
        b, sbn, sbk = (w_t, w_t.stride(0), 1) if w_t is not None else (w, 1, w.stride(0))
        check(lib.vitmi_gemm_colsum(_dt(dy2), T, K, N, _p(dy2), dy2.stride(0), 1, _p(b), sbn, sbk, _p(dx), K, _dt(dx),
                                    epi, None, None, 0, _p(aux) if aux is not None else None,
                                    K if aux is not None else 0, _p(cs), _p(ws), _stream()), "vitmi_gemm_colsum")
        return dx, cs
    if w_t is not None:
        gemm_raw(_dt(dy2), T, K, N, dy2, dy2.stride(0), 1, w_t, w_t.stride(0), 1, dx, K, _dt(dx), epi, None, None, 0,
                 aux, K if aux is not None else 0)
    else:
        gemm_raw(_dt(dy2), T, K, N, dy2, dy2.stride(0), 1, w, 1, w.stride(0), dx, K, _dt(dx), epi, None, None, 0, aux,
                 K if aux is not None else 0)
    return dx


def linear_bwd_w(dy2, x2):
    """dw[N,K] (fp32) = dy2[T,N]^T @ x2[T,K]."""
    T, N = dy2.shape
    K = x2.shape[1]
    dw = torch.empty((N, K), dtype=torch.float32, device=dy2.device)
    dt = _dt(dy2)
    ws = _ws(lib.vitmi_gemm_wgrad_workspace_bytes(dt, N, K, T), dy2.device)
    check(lib.vitmi_gemm_wgrad(dt, N, K, T, _p(dy2), dy2.stride(0), _p(x2), x2.stride(0), _p(dw), K, _p(ws),
                               _stream()), "vitmi_gemm_wgrad")
    return dw


def linear_bwd_w_pair(dy_a, x_a, dy_b, x_b):
    """The two weight gradients of a sub-block, dy_a^T @ x_a and dy_b^T @ x_b over the same tokens, through
    vitmi_gemm_wgrad_pair (one launch for bf16 operands whose extents are multiples of 256)."""
    T, Na = dy_a.shape
    Ka, Nb, Kb = x_a.shape[1], dy_b.shape[1], x_b.shape[1]
    assert dy_b.shape[0] == T and x_a.shape[0] == T and x_b.shape[0] == T
    dwa = torch.empty((Na, Ka), dtype=torch.float32, device=dy_a.device)
    dwb = torch.empty((Nb, Kb), dtype=torch.float32, device=dy_a.device)
    dt = _dt(dy_a)
    ws = _ws(lib.vitmi_gemm_wgrad_pair_workspace_bytes(dt, T, Na, Ka, Nb, Kb), dy_a.device)
    check(lib.vitmi_gemm_wgrad_pair(dt, T, Na, Ka, _p(dy_a), dy_a.stride(0), _p(x_a), x_a.stride(0), _p(dwa),
                                    Nb, Kb, _p(dy_b), dy_b.stride(0), _p(x_b), x_b.stride(0), _p(dwb), _p(ws),
                                    _stream()), "vitmi_gemm_wgrad_pair")
    return dwa, dwb


def weight_t(param, cdt):
    """Transposed bf16 copy for the input-gradient product, or None in fp32 mode."""
    return compute_copy_t(param, cdt) if cdt == torch.bfloat16 else None


def colsum(x2):
    rows, cols = x2.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2.device)
    ws = _ws(lib.vitmi_colsum_workspace_bytes(rows, cols), x2.device)
    check(lib.vitmi_colsum(_p(x2), _dt(x2), rows, cols, x2.stride(0), _p(out), _p(ws), _stream()), "vitmi_colsum")
    return out


def layernorm_fwd(x2, gamma, beta, eps):
    """x2: [rows, D] with unit column stride (row stride free).  Returns (y dense, mean, rstd)."""
    rows, D = x2.shape
    y = torch.empty((rows, D), dtype=x2.dtype, device=x2.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    check(lib.vitmi_layernorm_fwd(_p(x2), x2.stride(0), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _dt(x2), rows,
                                  D, eps, _stream()), "vitmi_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy2, x2, gamma, mean, rstd, dres=None, need_affine=True, want_dres_colsum=False):
    """Returns (dx, dgamma, dbeta) and, with want_dres_colsum, the fp32 column sums of dres as a 4th value."""
    rows, D = x2.shape
    dx = torch.empty((rows, D), dtype=x2.dtype, device=x2.device)
    dg = torch.empty(D, dtype=torch.float32, device=x2.device) if need_affine else None
    db = torch.empty(D, dtype=torch.float32, device=x2.device) if need_affine else None
    rs = torch.empty(D, dtype=torch.float32, device=x2.device) if want_dres_colsum else None
    ws = _ws(lib.vitmi_layernorm_bwd_workspace_bytes(rows, D), x2.device)
    check(lib.vitmi_layernorm_bwd(_p(dy2), _p(x2), x2.stride(0), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx),
                                  _p(dg), _p(db), _p(rs), _p(ws), _dt(x2), rows, D, _stream()), "vitmi_layernorm_bwd")
    if want_dres_colsum:
        return dx, dg, db, rs
    return dx, dg, db


QKV_HEAD_MAJOR = 1      # qkv stored as planes [3][H][B*N][DH] (what the whole-block forward leaves for bf16 heads of 64)


def attn_fwd(qkv, B, N, H, DH, want_probs, layout=0):
    I = H * DH
    out = torch.empty((B, N, I), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    probs = torch.empty((B, H, N, N), dtype=qkv.dtype, device=qkv.device) if want_probs else None
    check(lib.vitmi_attn_fwd(_p(qkv), _p(out), _p(lse), _p(probs), _dt(qkv), B, N, H, DH, layout, _stream()),
          "vitmi_attn_fwd")
    return out, lse, probs


def attn_bwd(qkv, out, dout, lse, B, N, H, DH, layout=0):
    """dqkv [B,N,3*H*DH] in the row layout of to_qkv's output; `layout` says how the saved qkv is stored."""
    dqkv = torch.empty_like(qkv)
    ws = _ws(lib.vitmi_attn_bwd_workspace_bytes(_dt(qkv), B, N, H, DH), qkv.device)
    check(lib.vitmi_attn_bwd(_p(qkv), _p(out), _p(dout), _p(lse), None, _p(dqkv), _p(ws), _dt(qkv), B, N, H, DH,
                             layout & QKV_HEAD_MAJOR, _stream()), "vitmi_attn_bwd")
    return dqkv


def attn_bwd_dprobs(qkv, lse, dprobs, B, N, H, DH):
    """The share of dqkv that a gradient on the exported probabilities contributes (q and k slots; v slots zero)."""
    dprobs = dprobs.contiguous()
    if dprobs.dtype != qkv.dtype:
        dprobs = cast(dprobs, qkv.dtype)
    dqkv = torch.empty_like(qkv)
    ws = _ws(lib.vitmi_attn_bwd_dprobs_workspace_bytes(_dt(qkv), B, N, H, DH), qkv.device)
    check(lib.vitmi_attn_bwd(_p(qkv), None, None, _p(lse), _p(dprobs), _p(dqkv), _p(ws), _dt(qkv), B, N, H, DH, 0,
                             _stream()), "vitmi_attn_bwd(dprobs)")
    return dqkv


class AttnProbsFn(torch.autograd.Function):
    """qkv [B,N,3*H*DH] -> softmax(q k^T * scale) [B,H,N,N] as a differentiable tensor (mymodel/ViT.py:96-100): the
    first half of the attention core when a caller wants the gradient that reaches the probabilities
    (vit_grad_rollout.py:38-56 hooks attn_drop forwards AND backwards)."""

    @staticmethod
    def forward(ctx, qkv, heads):
        _need_gpu(qkv)
        qkv = qkv.contiguous()
        B, N, I3 = qkv.shape
        dh = I3 // 3 // heads
        _, lse, probs = attn_fwd(qkv, B, N, heads, dh, True)
        ctx.save_for_backward(qkv, lse)
        ctx.geom = (B, N, heads, dh)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        qkv, lse = ctx.saved_tensors
        B, N, H, dh = ctx.geom
        return attn_bwd_dprobs(qkv, lse, dprobs, B, N, H, dh), None


class ProbsApplyFn(torch.autograd.Function):
    """(probs [B,H,N,N], qkv) -> probs @ v merged over heads, [B,N,H*DH] (mymodel/ViT.py:102-103): the second half.
    One small GEMM per (image, head) through strided views, forwards and for each of the two gradients; meant for
    attention-map analysis on a few images."""

    @staticmethod
    def forward(ctx, probs, qkv, heads):
        _need_gpu(probs, qkv)
        probs, qkv = probs.contiguous(), qkv.contiguous()
        B, N, I3 = qkv.shape
        I = I3 // 3
        dh = I // heads
        dt = _dt(qkv)
        o = torch.empty((B, N, I), dtype=qkv.dtype, device=qkv.device)
        for b in range(B):
            for h in range(heads):
                v = qkv[b, :, 2 * I + h * dh:]
                gemm_raw(dt, N, dh, N, probs[b, h], N, 1, v, 1, I3, o[b, :, h * dh:], I, dt)
        ctx.save_for_backward(probs, qkv)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        probs, qkv = ctx.saved_tensors
        heads = ctx.heads
        B, N, I3 = qkv.shape
        I = I3 // 3
        dh = I // heads
        dt = _dt(qkv)
        do = do.contiguous()
        if do.dtype != qkv.dtype:
            do = cast(do, qkv.dtype)
        dprobs = torch.empty_like(probs)
        dqkv = torch.zeros_like(qkv)
        for b in range(B):
            for h in range(heads):
                v = qkv[b, :, 2 * I + h * dh:]
                dob = do[b, :, h * dh:]
                gemm_raw(dt, N, N, dh, dob, I, 1, v, I3, 1, dprobs[b, h], N, dt)                 # dP = dO V^T
                gemm_raw(dt, N, dh, N, probs[b, h], 1, N, dob, 1, I, dqkv[b, :, 2 * I + h * dh:], I3, dt)   # dV = P^T dO
        return dprobs, dqkv, None


def add(a, b):
    y = torch.empty_like(a)
    check(lib.vitmi_add(_p(a), _p(b), _p(y), _dt(a), a.numel(), _stream()), "vitmi_add")
    return y


def scale_add_clip(x, scale=1.0, addend=None, lo=float("-inf"), hi=float("inf")):
    """clamp(x * scale + addend, lo, hi) elementwise; addend fp32 of x's shape or None."""
    _need_gpu(x, addend)
    x = x.contiguous()
    if addend is not None and (addend.dtype != torch.float32 or addend.numel() != x.numel() or not addend.is_contiguous()):
        raise ValueError("addend must be a contiguous float32 tensor of x's size")
    y = torch.empty_like(x)
    check(lib.vitmi_scale_add_clip(_p(x), _p(addend), _p(y), _dt(x), x.numel(), float(scale), float(lo), float(hi),
                                   _stream()), "vitmi_scale_add_clip")
    return y


def seq_ord(unk_mask, n, masked_row, device):
    """int64 [n]: i*(1-m[i]) + m[i]*masked_row from a 0/1 mask of n entries (any integer dtype / device)."""
    m = unk_mask.reshape(-1)
    if m.numel() != n:
        raise ValueError(f"unk_mask has {m.numel()} entries, the sequence has {n} tokens")
    if m.device != torch.device(device) or m.dtype not in (torch.int64, torch.int32):
        m = m.to(device=device, dtype=torch.long)
    m = m.contiguous()
    _need_gpu(m)
    out = torch.empty(n, dtype=torch.int64, device=m.device)
    check(lib.vitmi_seq_ord(_p(m), int(m.dtype == torch.int64), _p(out), n, int(masked_row), _stream()), "vitmi_seq_ord")
    return out


def jigsaw_gather(x, p, src, bank=None, bank_idx=None, fill=None):
    """Patch-level gather over an image batch (see vitmi_jigsaw_gather).  x [B,C,H,W]; src int32 [L] on the
    device; bank [P,C,H,W]; bank_idx int32 [B]; fill [B,C,p,p].  Returns a new [B,C,H,W] tensor."""
    _need_gpu(x, src, bank, bank_idx, fill)
    x = x.contiguous()
    B, C, H, W = x.shape
    if src.dtype != torch.int32 or src.numel() != (H // p) * (W // p):
        raise ValueError("src must be int32 with one entry per patch")
    for t in (bank, fill):
        if t is not None and (t.dtype != x.dtype or not t.is_contiguous()):
            raise ValueError("bank / fill must be contiguous and of the images' dtype")
    if bank is not None and (tuple(bank.shape[1:]) != (C, H, W) or bank_idx is None or bank_idx.dtype != torch.int32
                             or bank_idx.numel() != B):
        raise ValueError("bank must be [P,C,H,W] with an int32 bank_idx per image")
    if fill is not None and tuple(fill.shape) != (B, C, p, p):
        raise ValueError("fill must be [B,C,p,p]")
    out = torch.empty_like(x)
    check(lib.vitmi_jigsaw_gather(_p(x), _p(out), _dt(x), B, C, H, W, p, _p(src), _p(bank),
                                  bank.shape[0] if bank is not None else 0, _p(bank_idx), _p(fill), _stream()),
          "vitmi_jigsaw_gather")
    return out


def patch_weighted_mean(x, p, weight):
    """[B,C,p,p]: weighted mean of each image's patches (see vitmi_patch_weighted_mean); weight fp32 [L]."""
    _need_gpu(x, weight)
    x = x.contiguous()
    B, C, H, W = x.shape
    if weight.dtype != torch.float32 or weight.numel() != (H // p) * (W // p):
        raise ValueError("weight must be float32 with one entry per patch")
    out = torch.empty((B, C, p, p), dtype=x.dtype, device=x.device)
    check(lib.vitmi_patch_weighted_mean(_p(x), _dt(x), B, C, H, W, p, _p(weight), _p(out), _stream()),
          "vitmi_patch_weighted_mean")
    return out


# ---------------------------------------------------------------------------------------------
# per-sample gradient norms (DP-SGD): a backward pass run inside `per_sample_norms` adds every parameter's share
# of ||g_i||^2 to .sq[i] and hands autograd no parameter gradients; input gradients flow as usual
# ---------------------------------------------------------------------------------------------
_PS = None
_PS2 = None


class per_sample_norms:
    """keep=True: every node also keeps what `clipped_replay` needs to form sum_i c_i g_i afterwards without a second
    pass over the network (the upstream gradients of its parameter products and the per-image sums of its vector
    parameters)."""

    def __init__(self, batch, device, keep=False):
        self.batch = int(batch)
        self.keep = bool(keep)
        self.sq = torch.zeros(self.batch, dtype=torch.float32, device=device)

    def __enter__(self):
        global _PS
        if _PS is not None:
            raise RuntimeError("per_sample_norms does not nest")
        _PS = self
        return self

    def __exit__(self, *exc):
        global _PS
        _PS = None
        return False

    def seg(self, rows):
        if rows % self.batch:
            raise ValueError(f"{rows} rows do not split into {self.batch} samples")
        return rows // self.batch


class clipped_replay:
    """A backward pass inside hands every parameter sum_i coef[i] g_i, formed from what the norm pass
    (`per_sample_norms(keep=True)` over the same graph, retain_graph=True) kept: the upstream gradient of image i under
    the loss sum_i coef[i] loss_i is coef[i] times its value under sum_i loss_i, so the input-gradient chain is not run
    again -- each node scales the rows of one operand of its weight-gradient products by the image's factor and
    combines the per-image sums of its vector parameters.  Gradients with respect to the nodes' inputs are returned as
    zeros (stride-0 placeholders): only parameter gradients are meaningful after such a pass."""

    def __init__(self, coef):
        self.coef = coef.detach().to(torch.float32).contiguous()

    def __enter__(self):
        global _PS2
        if _PS2 is not None or _PS is not None:
            raise RuntimeError("clipped_replay does not nest")
        _PS2 = self
        return self

    def __exit__(self, *exc):
        global _PS2
        _PS2 = None
        return False


def _kept(ctx):
    k = getattr(ctx, "kept", None)
    if k is None:
        raise RuntimeError("clipped_replay: this node kept nothing -- run per_sample_norms(keep=True) over the same "
                           "graph first (retain_graph=True)")
    ctx.kept = None             # one replay per norm pass: the kept tensors are released here
    return k


def _no_grad_like(shape, dtype, device):
    """A zero gradient of the given shape that occupies one element (for inputs whose gradient a replay does not form)."""
    return torch.zeros((), dtype=dtype, device=device).expand(shape)


def seg_scale_rows(x2, coef, seg):
    """x2[t, :] * coef[t // seg] as a new tensor."""
    rows, cols = x2.shape
    out = torch.empty((rows, cols), dtype=x2.dtype, device=x2.device)
    check(lib.vitmi_seg_scale_rows(_p(x2), _dt(x2), rows, cols, x2.stride(0), int(seg), _p(coef), _p(out), cols, _stream()),
          "vitmi_seg_scale_rows")
    return out


def seg_combine(sums, coef):
    """sum_i coef[i] * sums[i, :] (fp32 [cols])."""
    n, cols = sums.shape
    out = torch.empty(cols, dtype=torch.float32, device=sums.device)
    check(lib.vitmi_seg_combine(_p(sums), _dt(sums), n, cols, sums.stride(0), _p(coef), _p(out), _stream()),
          "vitmi_seg_combine")
    return out


def ps_weight(ps, dy2, x2):
    """|| dy_i^T x_i ||_F^2 of y = x W^T, per sample."""
    rows, P = dy2.shape
    Q = x2.shape[1]
    seg = ps.seg(rows)
    if seg == 1:            # one row per sample: the gradient is an outer product, its norm a product of norms
        a = torch.zeros_like(ps.sq)
        b = torch.zeros_like(ps.sq)
        check(lib.vitmi_seg_sqnorm(_p(dy2), _dt(dy2), rows, P, dy2.stride(0), 1, _p(a), _stream()), "vitmi_seg_sqnorm")
        check(lib.vitmi_seg_sqnorm(_p(x2), _dt(x2), rows, Q, x2.stride(0), 1, _p(b), _stream()), "vitmi_seg_sqnorm")
        ps.sq.addcmul_(a, b)
        return
    dt = _dt(dy2)
    ws = _ws(lib.vitmi_wgrad_sqnorm_workspace_bytes(dt, P, Q, rows, seg), dy2.device)
    check(lib.vitmi_wgrad_sqnorm(dt, P, Q, rows, seg, _p(dy2), dy2.stride(0), _p(x2), x2.stride(0), _p(ps.sq), _p(ws),
                                 _stream()), "vitmi_wgrad_sqnorm")


def ps_bias(ps, dy2, want_sums=False):
    """Adds || sum_t dy_i ||^2; with want_sums returns the per-image column sums [B, cols] (fp32)."""
    rows, cols = dy2.shape
    sums = torch.empty((ps.batch, cols), dtype=torch.float32, device=dy2.device) if want_sums else None
    check(lib.vitmi_seg_colsum_sqnorm(_p(dy2), _dt(dy2), rows, cols, dy2.stride(0), ps.seg(rows), _p(ps.sq), _p(sums),
                                      _stream()), "vitmi_seg_colsum_sqnorm")
    return sums


def ps_layernorm(ps, dy2, x2, mean, rstd, want_sums=False):
    """LayerNorm weight and bias; with want_sums returns (per-image sums of dy, per-image sums of dy * xhat)."""
    rows, D = x2.shape
    dy2 = dy2.contiguous()
    sb = torch.empty((ps.batch, D), dtype=torch.float32, device=dy2.device) if want_sums else None
    sg = torch.empty((ps.batch, D), dtype=torch.float32, device=dy2.device) if want_sums else None
    check(lib.vitmi_seg_layernorm_sqnorm(_p(dy2), _p(x2), x2.stride(0), _p(mean), _p(rstd), _dt(x2), rows, D, ps.seg(rows),
                                         _p(ps.sq), _p(sb), _p(sg), _stream()), "vitmi_seg_layernorm_sqnorm")
    return (sb, sg) if want_sums else None


def ps_rows(ps, x2, cols, ld, rows, seg):
    check(lib.vitmi_seg_sqnorm(_p(x2), _dt(x2), rows, cols, ld, seg, _p(ps.sq), _stream()), "vitmi_seg_sqnorm")


def dropout(x, p, seed, out=None):
    """x * keep / (1 - p), the mask a function of (seed, element index); out=x works in place."""
    x = x.contiguous()
    y = torch.empty_like(x) if out is None else out
    check(lib.vitmi_dropout(_p(x), _p(y), _dt(x), x.numel(), float(p), int(seed), _stream()), "vitmi_dropout")
    return y


def new_seed():
    """A 63-bit seed from torch's CPU generator (so torch.manual_seed governs the dropout masks)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class DropoutFn(torch.autograd.Function):
    """nn.Dropout in training mode (mymodel/ViT.py:68,70,90); the backward regenerates the mask from the seed."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _need_gpu(x)
        ctx.p, ctx.seed = p, seed
        return dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


class AddFn(torch.autograd.Function):
    """Residual add (mymodel/ViT.py:119-120) as its own node, for the paths where it cannot ride on a GEMM epilogue."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        return add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


_block_ws_bytes = {}     # workspace sizes of the whole-block calls, by shape


def _as2d(x):
    """[..., D] -> [rows, D] view with unit column stride (copies only if the layout forbids a view)."""
    D = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1:
        return x
    if x.is_contiguous():
        return x.view(-1, D)
    return x.contiguous().view(-1, D)


# ---------------------------------------------------------------------------------------------
# autograd functions
# ---------------------------------------------------------------------------------------------
class PatchEmbedFn(torch.autograd.Function):
    """Conv2d(k=s=p)+bias then flatten/transpose (mymodel/ViT.py:33,43) -> [B, tokens, D].

    The patches are gathered in place by the GEMM's loads: multiples of 8 pixels as they are, the 4 x 4 patches of
    the 32-pixel geometry as groups of two adjacent patches against a block form of the weight
    (vitmi_patch_embed_fwd_wide), fp32 compute through the strided fp32 kernel.  Only geometries none of these
    covers (an image width that is not a multiple of 8 in bf16) go through a gathered [tokens, C*p*p] copy."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdt):
        _need_gpu(x, weight)
        B, C, H, W = x.shape
        D, _, p, _ = weight.shape
        per = (H // p) * (W // p)
        if x.dtype not in _DT:
            x = x.float()
        if cdt == torch.float32 and x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        ctx.x_dtype = x.dtype
        ctx.geom = (B, C, H, W, p, per)
        ctx.cdt = cdt
        ctx.has_bias = bias is not None
        wide = 0
        if cdt == torch.bfloat16 and not (p % 8 == 0 and W % 8 == 0):
            wide = int(lib.vitmi_patch_wide_group(_dt(cdt), _dt(cdt), B, C, H, W, p, D))
        ctx.wide = wide
        if wide:
            x = cast(x, torch.bfloat16)
            w_wide, b_wide = patch_wide_weight(weight, bias, wide)
            out = torch.empty((B, per, D), dtype=cdt, device=x.device)
            check(lib.vitmi_patch_embed_fwd_wide(_p(x), _p(w_wide), _p(b_wide), _p(out), B, C, H, W, p, D, per * D,
                                                 _stream()), "vitmi_patch_embed_fwd_wide")
            ctx.in_place = True
            ctx.save_for_backward(x, weight)
            return out
        w_c = compute_copy(weight, cdt).view(D, C * p * p)
        in_place = (p % 8 == 0 and W % 8 == 0) or cdt == torch.float32
        ctx.in_place = in_place
        if in_place:
            if cdt == torch.bfloat16 and x.dtype != torch.bfloat16:
                # one elementwise pass; the direct-to-LDS patch gather then reads 16-byte bf16 chunks
                # of the image in place (forward and weight gradient), and the saved copy is half the size
                x = cast(x, torch.bfloat16)
            out = torch.empty((B, per, D), dtype=cdt, device=x.device)
            bias_f = bias.detach().float() if bias is not None else None
            check(lib.vitmi_patch_embed_fwd(_p(x), _dt(x), _p(w_c), _p(bias_f), _p(out), _dt(cdt), B, C, H, W, p, D,
                                            per * D, _stream()), "vitmi_patch_embed_fwd")
            ctx.save_for_backward(x, weight)
            return out
        patches = torch.empty((B * per, C * p * p), dtype=cdt, device=x.device)
        check(lib.vitmi_patchify(_p(x), _dt(x), _p(patches), _dt(cdt), B, C, H, W, p, _stream()), "vitmi_patchify")
        out, _ = linear_fwd(patches, w_c, bias.detach().float() if bias is not None else None)
        ctx.save_for_backward(patches, weight)
        return out.view(B, per, D)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, C, H, W, p, per = ctx.geom
        D = weight.shape[0]
        rp = _PS2
        if rp is not None:                  # dW = sum_i c_i dy_i^T P_i: scale the rows of the kept dy, then as usual
            dy = seg_scale_rows(_kept(ctx).view(B * per, D), rp.coef, per).view(B, per, D)
            if ctx.needs_input_grad[0]:
                raise NotImplementedError("clipped replay with a gradient into the images")
        dy = dy.contiguous()
        ps = _PS
        if ps is not None:
            if ps.keep:
                ctx.kept = dy
            if ctx.in_place:
                patches = torch.empty((B * per, C * p * p), dtype=ctx.cdt, device=x.device)
                check(lib.vitmi_patchify(_p(x), _dt(x), _p(patches), _dt(ctx.cdt), B, C, H, W, p, _stream()),
                      "vitmi_patchify")
            else:
                patches = x
            ps_weight(ps, dy.view(B * per, D), patches)
            if ctx.has_bias:
                ps_bias(ps, dy.view(B * per, D))
            if ctx.needs_input_grad[0]:
                raise NotImplementedError("per-sample norms with a gradient into the images")
            return None, None, None, None
        if ctx.wide:
            dw = torch.empty((D, C * p * p), dtype=torch.float32, device=x.device)
            ws = _ws(lib.vitmi_patch_embed_bwd_w_wide_workspace_bytes(B, C, H, W, p, D), x.device)
            check(lib.vitmi_patch_embed_bwd_w_wide(_p(x), _p(dy), _p(dw), _p(ws), B, C, H, W, p, D, _stream()),
                  "vitmi_patch_embed_bwd_w_wide")
        elif ctx.in_place:
            dw = torch.empty((D, C * p * p), dtype=torch.float32, device=x.device)
            ws = _ws(lib.vitmi_patch_embed_bwd_w_workspace_bytes(_dt(dy), B, C, H, W, p, D), x.device)
            check(lib.vitmi_patch_embed_bwd_w(_p(x), _dt(x), _p(dy), per * D, _p(dw), _p(ws), _dt(dy), B, C, H, W, p,
                                              D, _stream()), "vitmi_patch_embed_bwd_w")
        else:
            dw = linear_bwd_w(dy.view(B * per, D), x)                # x holds the gathered patches
        db = colsum(dy.view(B * per, D)) if ctx.has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            w_c = compute_copy(weight, ctx.cdt).view(D, C * p * p)
            dpatch = linear_bwd_x(dy.view(B * per, D), w_c)          # [B*per, C*p*p]
            gh, gw = H // p, W // p
            dx = dpatch.view(B, gh, gw, C, p, p).permute(0, 3, 1, 4, 2, 5).reshape(B, C, H, W).to(ctx.x_dtype)
        return dx, dw.view_as(weight).to(weight.dtype), (db.to(weight.dtype) if db is not None else None), None


class TokensFn(torch.autograd.Function):
    """cls prepend + positional-embedding add, rows of the table chosen by seq_ord
    (mymodel/ViT.py:155-159, ViT_mask.py:156-172)."""

    @staticmethod
    def forward(ctx, patch, cls, pe2d, seq_ord):
        _need_gpu(patch, cls, pe2d, seq_ord)
        B, n1, D = patch.shape
        N = n1 + 1
        patch = patch.contiguous()
        cls_f = cls.detach().reshape(D).float().contiguous()
        pe_f = pe2d.detach().float().contiguous() if pe2d is not None else None
        out = torch.empty((B, N, D), dtype=patch.dtype, device=patch.device)
        check(lib.vitmi_tokens_assemble_fwd(_p(patch), _p(cls_f), _p(pe_f), _p(seq_ord), _p(out), _dt(patch), B, N, D,
                                            _stream()), "vitmi_tokens_assemble_fwd")
        ctx.seq_ord = seq_ord
        ctx.pe_shape = None if pe2d is None else tuple(pe2d.shape)
        ctx.cls_shape = tuple(cls.shape)
        ctx.pdtypes = (cls.dtype, None if pe2d is None else pe2d.dtype)
        return out

    @staticmethod
    def backward(ctx, dx):
        rp = _PS2
        B, N, D = dx.shape
        if rp is not None:
            dx = seg_scale_rows(_kept(ctx).view(B, N * D), rp.coef, 1).view(B, N, D)
        dx = dx.contiguous()
        ps = _PS
        if ps is not None:
            if ps.keep:
                ctx.kept = dx
            ps_rows(ps, dx, D, N * D, B, 1)                             # class token: row 0 of every image
            if ctx.pe_shape is not None:
                if ctx.seq_ord is not None:
                    raise NotImplementedError("per-sample norms with a masked positional gather")
                ps_rows(ps, dx, D, D, B * N, N)                         # table row r takes dx[i, r]
            return dx[:, 1:, :], None, None, None
        dcls = torch.empty(D, dtype=torch.float32, device=dx.device)
        rows = ctx.pe_shape[0] if ctx.pe_shape is not None else 0
        dpe = torch.empty((rows, D), dtype=torch.float32, device=dx.device) if rows else None
        ws = _ws(lib.vitmi_tokens_assemble_bwd_workspace_bytes(N, D), dx.device)
        check(lib.vitmi_tokens_assemble_bwd(_p(dx), _p(ctx.seq_ord), _p(dcls), _p(dpe), rows, _p(ws), _dt(dx), B, N,
                                            D, _stream()), "vitmi_tokens_assemble_bwd")
        return (dx[:, 1:, :], dcls.view(ctx.cls_shape).to(ctx.pdtypes[0]),
                dpe.to(ctx.pdtypes[1]) if dpe is not None else None, None)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim (mymodel/ViT.py:55,147)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_gpu(x, gamma)
        x2 = _as2d(x)
        g = gamma.detach().float()
        b = beta.detach().float()
        y, mean, rstd = layernorm_fwd(x2, g, b, eps)
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.shape = x.shape
        ctx.pdtype = gamma.dtype
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        rp = _PS2
        if rp is not None:                  # stand-alone LayerNorm (the head's): scale dy's rows, then the usual sums
            dys = seg_scale_rows(_kept(ctx), rp.coef, x2.shape[0] // rp.coef.numel())
            _, dg, db = layernorm_bwd(dys, x2, g, mean, rstd)
            return _no_grad_like(ctx.shape, dy.dtype, dy.device), dg.to(ctx.pdtype), db.to(ctx.pdtype), None
        ps = _PS
        if ps is not None:
            dy2 = _as2d(dy.contiguous())
            if ps.keep:
                ctx.kept = dy2
            ps_layernorm(ps, dy2, x2, mean, rstd)
            dx, _, _ = layernorm_bwd(dy2, x2, g, mean, rstd, need_affine=False)
            return dx.view(ctx.shape), None, None, None
        dx, dg, db = layernorm_bwd(_as2d(dy.contiguous()), x2, g, mean, rstd)
        return dx.view(ctx.shape), dg.to(ctx.pdtype), db.to(ctx.pdtype), None


class LinearFn(torch.autograd.Function):
    """nn.Linear (mymodel/ViT.py:148 head; also the stand-alone projections)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdt, out_dtype):
        _need_gpu(x, weight)
        x2 = _as2d(x)
        if x2.dtype != cdt:
            x2 = cast(x2, cdt)
        w_c = compute_copy(weight, cdt)
        bias_f = bias.detach().float() if bias is not None else None
        y, _ = linear_fwd(x2, w_c, bias_f, out_dtype=out_dtype)
        ctx.save_for_backward(x2, weight)
        ctx.cdt = cdt
        ctx.in_shape = x.shape
        ctx.in_dtype = x.dtype
        ctx.has_bias = bias is not None
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        rp = _PS2
        if rp is not None:
            dys = seg_scale_rows(_kept(ctx), rp.coef, x2.shape[0] // rp.coef.numel())
            dw = linear_bwd_w(dys, x2).to(weight.dtype)
            db = colsum(dys).to(weight.dtype) if ctx.has_bias else None
            return _no_grad_like(ctx.in_shape, ctx.in_dtype, dy.device), dw, db, None, None
        dy2 = _as2d(dy.contiguous())
        if dy2.dtype != ctx.cdt:
            dy2 = cast(dy2, ctx.cdt)
        w_c = compute_copy(weight, ctx.cdt)
        dx = linear_bwd_x(dy2, w_c, w_t=weight_t(weight, ctx.cdt)).view(ctx.in_shape)
        if dx.dtype != ctx.in_dtype:
            dx = cast(dx, ctx.in_dtype)
        ps = _PS
        if ps is not None:
            if ps.keep:
                ctx.kept = dy2
            ps_weight(ps, dy2, x2)
            if ctx.has_bias:
                ps_bias(ps, dy2)
            return dx, None, None, None, None
        dw = linear_bwd_w(dy2, x2).to(weight.dtype)
        db = colsum(dy2).to(weight.dtype) if ctx.has_bias else None
        return dx, dw, db, None, None


# ---------------------------------------------------------------------------------------------
# weight gradients of the small geometry, gathered after the backward pass
# ---------------------------------------------------------------------------------------------
# At embedding widths up to 512 (the 32-pixel models, ViT-S/16) a sub-block's two weight-gradient products are not
# launched by the sub-block's backward call: they are queued, and one vitmi_gemm_wgrad_multi call at
# the end of the autograd pass (an engine callback) forms all of them, 16 products per launch.  The gradient tensors
# handed to autograd in between are allocated but not yet filled, so the queue is used only when nothing reads them
# before the pass ends: the parameter has no gradient to accumulate into and no hooks (a GradReducer's buckets, a
# tensor hook).  VITMI_DEFER_WGRAD=0 or set_defer_wgrad(False) switches it off.
import ctypes as _ct


class _WgradJob(_ct.Structure):
    _fields_ = [("U", _ct.c_void_p), ("ldu", _ct.c_longlong), ("V", _ct.c_void_p), ("ldv", _ct.c_longlong),
                ("C", _ct.c_void_p), ("ldc", _ct.c_longlong), ("P", _ct.c_int), ("Q", _ct.c_int)]


class _SumsJob(_ct.Structure):
    _fields_ = [("partial", _ct.c_void_p), ("strips", _ct.c_int), ("M", _ct.c_int), ("D", _ct.c_int),
                ("db1", _ct.c_void_p), ("dgamma", _ct.c_void_p), ("dbeta", _ct.c_void_p), ("db2", _ct.c_void_p)]


_defer_wgrad = os.environ.get("VITMI_DEFER_WGRAD", "1") != "0"
# the vector gradients (biases, LayerNorm parameters) of the one-launch sub-block kernels ride on the same queue: their
# per-strip partial sums stay in the kept workspace and one vitmi_small_sums_multi launch reduces all of them
_defer_sums = os.environ.get("VITMI_DEFER_SUMS", "1") != "0"
# widths whose weight matrices are a few 128 x 128 tiles each (ViT-S/16 and below): gathered, they fill the chip
# without token slabs; from ViT-B/16 on the paired 256 x 256 launches of the sub-block calls are the faster form
_DEFER_MAX_WIDTH = 512
_deferred = {}      # graph task id -> {"groups": {(device, rows, stream): [jobs]}, "keep": [tensors]}
deferred_stats = {"jobs": 0, "flushes": 0}     # running totals, for tests and diagnostics
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)


def set_defer_wgrad(on):
    global _defer_wgrad
    _defer_wgrad = bool(on)


def set_defer_sums(on):
    global _defer_sums
    _defer_sums = bool(on)


def _defer_sums_ok(params):
    """The gathered vector gradients need what the gathered weight gradients need of their parameters, and fp32
    parameters: the tensors handed to autograd are the ones the late launch fills."""
    if not _defer_sums:
        return False
    for p in params:
        if p is None:
            continue
        if (p.dtype != torch.float32 or not p.is_leaf or p.grad is not None or p._backward_hooks
                or getattr(p, "_post_accumulate_grad_hooks", None)):
            return False
    return True


def _defer_ok(cdt, rows, D, params):
    if not _defer_wgrad or cdt != torch.bfloat16 or D > _DEFER_MAX_WIDTH or _graph_task_id is None:
        return False
    if _graph_task_id() < 0:
        return False
    for p in params:
        if (not p.is_leaf or p.grad is not None or p._backward_hooks
                or getattr(p, "_post_accumulate_grad_hooks", None)):
            return False
    return True


def _defer_sum_job(device, partial, strips, M, D, outs, params, keep):
    """outs / params: (db1, dgamma, dbeta, db2) result tensors (None: absent) and the parameters they belong to; the
    running pass's queue must exist already (the weight jobs of the same call open it)."""
    st = _deferred[_graph_task_id()]
    st.setdefault("sums", {}).setdefault((device, _stream()), []).append((partial, strips, M, D,
                                                                         tuple(None if t is None else _p(t) for t in outs),
                                                                         tuple(None if t is None else tuple(t.shape) for t in outs),
                                                                         params,
                                                                         tuple(None if t is None else weakref.ref(t) for t in outs)))
    st["keep"].extend(keep)
    deferred_stats["sums"] = deferred_stats.get("sums", 0) + 1


def _defer_jobs(rows, device, jobs, keep):
    """jobs: (U ptr, ldu, V ptr, ldv, C tensor, P, Q, parameter) over `rows` tokens; keep: tensors the operands live in.
    The result tensor itself is NOT kept: a second reference would make autograd's AccumulateGrad copy it (still
    empty) instead of adopting it as the parameter's .grad.  Only a weak reference is: a pass that does not want the
    gradient (torch.autograd.grad / backward(inputs=...) without the parameter) drops the tensor, its memory may serve
    another tensor by the time the queue is flushed, and the job must then not run.  The queue belongs to the running
    autograd pass (a pass started from inside another one, as reentrant checkpointing does, has its own)."""
    jobs = [(U, ldu, V, ldv, _p(C), P, Q, prm, weakref.ref(C)) for (U, ldu, V, ldv, C, P, Q, prm) in jobs]
    task = _graph_task_id()
    st = _deferred.get(task)
    if st is None:
        # a pass that ended without its callback (an exception) leaves jobs that will never be wanted
        for old in [t for t in _deferred if t < task - 64]:
            del _deferred[old]
        st = _deferred[task] = {"groups": {}, "keep": []}
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_deferred(task))
    st["groups"].setdefault((device, rows, _stream()), []).extend(jobs)
    st["keep"].extend(keep)
    deferred_stats["jobs"] += len(jobs)


def _flush_deferred(task):
    st = _deferred.pop(task, None)
    if st is None:
        return
    deferred_stats["flushes"] += 1
    for (device, rows, stream), jobs in st["groups"].items():
        wanted = []
        for job in jobs:
            if job[7].grad is None and job[8]() is None:   # nobody holds the result: the pass did not ask for it
                deferred_stats["dropped"] = deferred_stats.get("dropped", 0) + 1
            else:
                wanted.append(job[:8])
        jobs = wanted
        if not jobs:
            continue
        arr = (_WgradJob * len(jobs))()
        for a, (U, ldu, V, ldv, C, P, Q, param) in zip(arr, jobs):
            g = param.grad
            if g is not None and g.data_ptr() != C:
                # autograd did not adopt the tensor handed to it but copied it (its memory may be gone): the
                # product goes where the gradient now lives
                if g.dtype != torch.float32 or not g.is_contiguous() or tuple(g.shape) != (P, Q):
                    raise RuntimeError("deferred weight gradient: the parameter's .grad is not the fp32 [P, Q] tensor "
                                       "the backward pass handed to autograd (set_defer_wgrad(False) for this setup)")
                C = g.data_ptr()
            a.U, a.ldu, a.V, a.ldv, a.C, a.ldc, a.P, a.Q = U, ldu, V, ldv, C, Q, P, Q
        with torch.cuda.device(device):
            ws = _ws(lib.vitmi_gemm_wgrad_multi_workspace_bytes(_ct.addressof(arr), len(jobs), rows), device)
            check(lib.vitmi_gemm_wgrad_multi(_ct.addressof(arr), len(jobs), rows, _p(ws), stream), "vitmi_gemm_wgrad_multi")
    for (device, stream), jobs in st.get("sums", {}).items():
        arr = (_SumsJob * len(jobs))()
        spare = []
        for a, (partial, strips, M, D, ptrs, shapes, params, refs) in zip(arr, jobs):
            ptrs = list(ptrs)
            for i, prm in enumerate(params):
                if prm is None or ptrs[i] is None:
                    continue
                g = prm.grad
                if g is None and refs[i]() is None:
                    # nobody holds this result (the pass did not ask for it) and its memory may be someone else's by
                    # now: the sum goes to a throw-away tensor
                    spare.append(torch.empty(shapes[i], dtype=torch.float32, device=device))
                    ptrs[i] = spare[-1].data_ptr()
                    deferred_stats["dropped"] = deferred_stats.get("dropped", 0) + 1
                    continue
                if g is not None and g.data_ptr() != ptrs[i]:      # copied, not adopted: fill the copy
                    if g.dtype != torch.float32 or not g.is_contiguous() or tuple(g.shape) != shapes[i]:
                        raise RuntimeError("deferred vector gradient: the parameter's .grad is not the fp32 tensor the "
                                           "backward pass handed to autograd (set_defer_sums(False) for this setup)")
                    ptrs[i] = g.data_ptr()
            a.partial, a.strips, a.M, a.D = partial, strips, M, D
            a.db1, a.dgamma, a.dbeta, a.db2 = ptrs
        with torch.cuda.device(device):
            check(lib.vitmi_small_sums_multi(_ct.addressof(arr), len(jobs), stream), "vitmi_small_sums_multi")


class _BlockFn(torch.autograd.Function):
    """Base of the two sub-block nodes.  forward() always runs with grad mode off and ctx.needs_input_grad only
    mirrors requires_grad of the inputs, so whether a graph is being built is sampled where apply() is called."""
    _grad_mode = True

    @classmethod
    def apply(cls, *args):
        _BlockFn._grad_mode = torch.is_grad_enabled()
        return super().apply(*args)

    @staticmethod
    def builds_no_graph(ctx):
        return not (_BlockFn._grad_mode and any(ctx.needs_input_grad))


class AttnBlockFn(_BlockFn):
    """[LayerNorm ->] to_qkv (+ optional bias, tim/models/vision_transformer.py:74) ->
    softmax(q k^T * scale) v -> [to_out] [+ x]  (mymodel/ViT.py:58, 93-104, 119) as one autograd node.

    Returns (y, probs); probs is the [B,H,N,N] post-softmax tensor when export is requested
    (the tensor the reference exposes to forward hooks on Attention.attend), else None.
    """

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_qkv, w_out, b_out, heads, eps, add_residual, want_probs, cdt, b_qkv=None):
        _need_gpu(x, w_qkv)
        B, N, D = x.shape
        x2 = _as2d(x)
        if x2.dtype != cdt:
            x2 = cast(x2, cdt)
        inner = w_qkv.shape[0] // 3
        dh = inner // heads
        whole = ln_w is not None and add_residual and w_out is not None and x2.is_contiguous()
        if (whole and _infer_fusion and cdt == torch.bfloat16 and _BlockFn.builds_no_graph(ctx) and D % 8 == 0
                and inner % 8 == 0):
            # no autograd graph is being built: LayerNorm folded into the qkv product, nothing saved
            wg, sq, cq = folded_linear(w_qkv, ln_w, ln_b, b_qkv)
            dev = x2.device
            o_probs = torch.empty((B, heads, N, N), dtype=cdt, device=dev) if want_probs else None
            y = torch.empty((B * N, D), dtype=cdt, device=dev)
            key = ("attn_infer", B, N, D, heads, dh)
            nbytes = _block_ws_bytes.get(key)
            if nbytes is None:
                nbytes = _block_ws_bytes[key] = int(lib.vitmi_attn_block_infer_workspace_bytes(B, N, D, heads, dh))
            ws = _ws(nbytes, dev)
            check(lib.vitmi_attn_block_infer(B, N, D, heads, dh, _p(x2), eps, _p(wg), _p(cq), _p(sq),
                                             _p(compute_copy(w_out, cdt)),
                                             _p(b_out.detach().float()) if b_out is not None else None, _p(o_probs),
                                             _p(y), _p(ws), _stream()), "vitmi_attn_block_infer")
            if o_probs is not None:
                ctx.mark_non_differentiable(o_probs)
            return y.view(B, N, -1), o_probs
        if whole:                       # the usual pre-norm residual block: one foreign call for the four launches
            g = ln_w.detach().float()
            rows, dev, f32 = B * N, x2.device, torch.float32
            xn = torch.empty((rows, D), dtype=cdt, device=dev)
            mean, rstd = torch.empty(rows, dtype=f32, device=dev), torch.empty(rows, dtype=f32, device=dev)
            qkv = torch.empty((rows, 3 * inner), dtype=cdt, device=dev)
            o = torch.empty((B, N, inner), dtype=cdt, device=dev)
            lse = torch.empty((B, heads, N), dtype=f32, device=dev)
            probs = torch.empty((B, heads, N, N), dtype=cdt, device=dev) if want_probs else None
            y = torch.empty((rows, D), dtype=cdt, device=dev)
            qkv_layout = int(lib.vitmi_attn_block_qkv_layout(_dt(cdt), B, N, D, heads, dh))
            check(lib.vitmi_attn_block_fwd(_dt(cdt), B, N, D, heads, dh, _p(x2), _p(g), _p(ln_b.detach().float()), eps,
                                           _p(compute_copy(w_qkv, cdt)),
                                           _p(b_qkv.detach().float()) if b_qkv is not None else None,
                                           _p(compute_copy(w_out, cdt)),
                                           _p(b_out.detach().float()) if b_out is not None else None, _p(xn), _p(mean),
                                           _p(rstd), _p(qkv), _p(o), _p(lse), _p(probs), _p(y), qkv_layout, _stream()),
                  "vitmi_attn_block_fwd")
        else:
            if ln_w is not None:
                g = ln_w.detach().float()
                xn, mean, rstd = layernorm_fwd(x2, g, ln_b.detach().float(), eps)
            else:
                g = mean = rstd = None
                xn = x2
            wq = compute_copy(w_qkv, cdt)
            qkv, _ = linear_fwd(xn, wq, b_qkv.detach().float() if b_qkv is not None else None)
            o, lse, probs = attn_fwd(qkv, B, N, heads, dh, want_probs)
            o2 = o.view(B * N, inner)
            if w_out is not None:
                wo = compute_copy(w_out, cdt)
                y, _ = linear_fwd(o2, wo, b_out.detach().float() if b_out is not None else None,
                                  res=x2 if add_residual else None)
            else:
                y = add(o2, x2) if add_residual else o2
        ctx.whole = whole
        ctx.qkv_layout = qkv_layout if whole else 0      # how the saved qkv is stored (planes for bf16 heads of 64)
        ctx.save_for_backward(x2, g, mean, rstd, xn if ln_w is not None else None, qkv, o, lse, w_qkv, w_out)
        ctx.meta = (B, N, D, heads, dh, add_residual, cdt, ln_w is not None, b_out is not None,
                    ln_w.dtype if ln_w is not None else None, x.dtype)
        ctx.bqkv_dtype = b_qkv.dtype if b_qkv is not None else None
        ctx.vec_params = (None, ln_w, ln_b, b_out)       # (db1, dgamma, dbeta, db2) slots of the gathered vector sums
        if probs is not None:
            ctx.mark_non_differentiable(probs)
        return y.view(B, N, -1), probs

    @staticmethod
    def backward(ctx, dy, _dprobs):
        x2, g, mean, rstd, xn, qkv, o, lse, w_qkv, w_out = ctx.saved_tensors
        B, N, D, heads, dh, add_residual, cdt, has_ln, has_bo, ln_dtype, in_dtype = ctx.meta
        inner = heads * dh
        rp = _PS2
        if rp is not None:
            # sum_i c_i g_i from the norm pass's tensors: rows of the 'narrow' operand of each product scaled by c
            dy2, dqkv2, s_bq, s_ln = _kept(ctx)
            xn_ = xn if xn is not None else x2
            xs = seg_scale_rows(xn_, rp.coef, N)
            dwo = dbo = dbq = dg = db = None
            if w_out is not None:
                dys = seg_scale_rows(dy2, rp.coef, N)
                dwo, dwq = linear_bwd_w_pair(dys, o.view(B * N, inner), dqkv2, xs)
                dwo = dwo.to(w_out.dtype)
                if has_bo:
                    dbo = colsum(dys).to(w_out.dtype)
            else:
                dwq = linear_bwd_w(dqkv2, xs)
            if s_bq is not None:
                dbq = seg_combine(s_bq, rp.coef).to(ctx.bqkv_dtype)
            if has_ln:
                db, dg = seg_combine(s_ln[0], rp.coef).to(ln_dtype), seg_combine(s_ln[1], rp.coef).to(ln_dtype)
            return (_no_grad_like((B, N, D), in_dtype, dy.device), dg, db, dwq.to(w_qkv.dtype), dwo, dbo, None, None, None,
                    None, None, dbq)
        dy2 = _as2d(dy.contiguous())
        if dy2.dtype != cdt:
            dy2 = cast(dy2, cdt)
        ps = _PS
        if ps is not None:
            xn_ = xn if xn is not None else x2
            o2 = o.view(B * N, inner)
            if w_out is not None:
                ps_weight(ps, dy2, o2)
                if has_bo:
                    ps_bias(ps, dy2)
                do = linear_bwd_x(dy2, compute_copy(w_out, cdt), w_t=weight_t(w_out, cdt))
            else:
                do = dy2
            dqkv2 = attn_bwd(qkv, o, do.view(B, N, inner), lse, B, N, heads, dh, ctx.qkv_layout).view(B * N, 3 * inner)
            ps_weight(ps, dqkv2, xn_)
            s_bq = s_ln = None
            if ctx.bqkv_dtype is not None:
                s_bq = ps_bias(ps, dqkv2, want_sums=ps.keep)
            dxn = linear_bwd_x(dqkv2, compute_copy(w_qkv, cdt), w_t=weight_t(w_qkv, cdt))
            if has_ln:
                s_ln = ps_layernorm(ps, dxn, x2, mean, rstd, want_sums=ps.keep)
                dx, _, _ = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2 if add_residual else None, need_affine=False)
            else:
                dx = add(dxn, dy2) if add_residual else dxn
            if ps.keep:
                ctx.kept = (dy2, dqkv2, s_bq, s_ln)
            dx = dx.view(B, N, D)
            if dx.dtype != in_dtype:
                dx = cast(dx, in_dtype)
            return (dx,) + (None,) * 11
        if ctx.whole and dy2.is_contiguous():
            rows, dev, f32 = B * N, x2.device, torch.float32
            dx = torch.empty((rows, D), dtype=cdt, device=dev)
            dg, db = torch.empty(D, dtype=f32, device=dev), torch.empty(D, dtype=f32, device=dev)
            dwq = torch.empty((3 * inner, D), dtype=f32, device=dev)
            dbq = torch.empty(3 * inner, dtype=f32, device=dev) if ctx.bqkv_dtype is not None else None
            dwo = torch.empty((D, inner), dtype=f32, device=dev)
            dbo = torch.empty(D, dtype=f32, device=dev) if has_bo else None
            key = ("attn", _dt(cdt), B, N, D, heads, dh)
            nbytes = _block_ws_bytes.get(key)
            if nbytes is None:
                nbytes = _block_ws_bytes[key] = int(lib.vitmi_attn_block_bwd_workspace_bytes(_dt(cdt), B, N, D, heads, dh))
            ws = _ws(nbytes, dev)
            defer = D % 8 == 0 and inner % 8 == 0 and _defer_ok(cdt, rows, D, (w_qkv, w_out))
            strips = 0
            if (defer and dbq is None and not ctx.qkv_layout and ln_dtype == f32
                    and (dbo is None or w_out.dtype == f32) and _defer_sums_ok(ctx.vec_params)):
                strips = int(lib.vitmi_attn_block_bwd_sums_strips(_dt(cdt), B, N, D, heads, dh))
            late = strips > 0                # the vector sums ride on the queue as well
            check(lib.vitmi_attn_block_bwd(_dt(cdt), B, N, D, heads, dh, _p(dy2), _p(x2), _p(xn), _p(qkv), _p(o), _p(lse),
                                           _p(g), _p(mean), _p(rstd), _p(compute_copy(w_qkv, cdt)),
                                           _p(weight_t(w_qkv, cdt)), _p(compute_copy(w_out, cdt)),
                                           _p(weight_t(w_out, cdt)), _p(dx), None if late else _p(dg),
                                           None if late else _p(db), None if defer else _p(dwq), _p(dbq),
                                           None if defer else _p(dwo), None if late else _p(dbo), _p(ws), ctx.qkv_layout,
                                           _stream()), "vitmi_attn_block_bwd")
            if defer:
                dqkv = ws.data_ptr() + lib.vitmi_attn_block_bwd_dqkv_offset(_dt(cdt), B, N, D, heads, dh)
                _defer_jobs(rows, dev, [(_p(dy2), D, _p(o), inner, dwo, D, inner, w_out),
                                        (dqkv, 3 * inner, _p(xn), D, dwq, 3 * inner, D, w_qkv)], (dy2, o, xn, ws))
            if late:
                part = ws.data_ptr() + lib.vitmi_attn_block_bwd_sums_offset(_dt(cdt), B, N, D, heads, dh)
                _defer_sum_job(dev, part, strips, 0, D, (None, dg, db, dbo), ctx.vec_params, (ws,))
            dx = dx.view(B, N, D)
            if dx.dtype != in_dtype:
                dx = cast(dx, in_dtype)
            return (dx, dg.to(ln_dtype), db.to(ln_dtype), dwq.to(w_qkv.dtype), dwo.to(w_out.dtype),
                    dbo.to(w_out.dtype) if dbo is not None else None, None, None, None, None, None,
                    dbq.to(ctx.bqkv_dtype) if dbq is not None else None)
        if xn is None:
            xn = x2
        o2 = o.view(B * N, inner)
        dwo = dbo = None
        if w_out is not None:
            wo = compute_copy(w_out, cdt)
            dwo = linear_bwd_w(dy2, o2).to(w_out.dtype)
            fuse_dbo = has_bo and has_ln and add_residual     # colsum(dy) rides on the LayerNorm backward
            dbo = colsum(dy2).to(w_out.dtype) if (has_bo and not fuse_dbo) else None
            do = linear_bwd_x(dy2, wo, w_t=weight_t(w_out, cdt))
        else:
            do = dy2
        dqkv = attn_bwd(qkv, o, do.view(B, N, inner), lse, B, N, heads, dh, ctx.qkv_layout)
        dqkv2 = dqkv.view(B * N, 3 * inner)
        wq = compute_copy(w_qkv, cdt)
        dwq = linear_bwd_w(dqkv2, xn).to(w_qkv.dtype)
        dbq = colsum(dqkv2).to(ctx.bqkv_dtype) if ctx.bqkv_dtype is not None else None
        dxn = linear_bwd_x(dqkv2, wq, w_t=weight_t(w_qkv, cdt))
        dg = db = None
        if has_ln:
            if w_out is not None and fuse_dbo:
                dx, dg, db, rs = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2, want_dres_colsum=True)
                dbo = rs.to(w_out.dtype)
            else:
                dx, dg, db = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2 if add_residual else None)
            dg, db = dg.to(ln_dtype), db.to(ln_dtype)
        else:
            dx = add(dxn, dy2) if add_residual else dxn
        dx = dx.view(B, N, D)
        if dx.dtype != in_dtype:
            dx = cast(dx, in_dtype)
        return dx, dg, db, dwq, dwo, dbo, None, None, None, None, None, dbq


class FFBlockFn(_BlockFn):
    """[LayerNorm ->] Linear -> GELU(erf) -> Linear [+ x]  (mymodel/ViT.py:58, 65-73, 120)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, add_residual, cdt, drop=None):
        # drop = (p, seed_hidden, seed_out): nn.Dropout after GELU and after the second layer (mymodel/ViT.py:68,70)
        _need_gpu(x, w1)
        shape = x.shape
        x2 = _as2d(x)
        if x2.dtype != cdt:
            x2 = cast(x2, cdt)
        whole = ln_w is not None and add_residual and x2.is_contiguous() and drop is None
        if (whole and _infer_fusion and cdt == torch.bfloat16 and _BlockFn.builds_no_graph(ctx)
                and x2.shape[1] % 8 == 0 and w1.shape[0] % 8 == 0 and not lib.vitmi_get_gelu_mode()):
            # no autograd graph is being built: LayerNorm folded into the first product, no GELU derivative stored
            wg, s1, c1 = folded_linear(w1, ln_w, ln_b, b1)
            rows, D = x2.shape
            M = w1.shape[0]
            y = torch.empty((rows, D), dtype=cdt, device=x2.device)
            key = ("ff_infer", rows, D, M)
            nbytes = _block_ws_bytes.get(key)
            if nbytes is None:
                nbytes = _block_ws_bytes[key] = int(lib.vitmi_ff_block_infer_workspace_bytes(rows, D, M))
            ws = _ws(nbytes, x2.device)
            check(lib.vitmi_ff_block_infer(rows, D, M, _p(x2), eps, _p(wg), _p(c1), _p(s1), _p(compute_copy(w2, cdt)),
                                           _p(b2.detach().float()), _p(y), _p(ws), _stream()), "vitmi_ff_block_infer")
            return y.view(*shape[:-1], w2.shape[0])
        if whole:                       # the usual pre-norm residual block: one foreign call for the three launches
            g = ln_w.detach().float()
            rows, D = x2.shape
            M = w1.shape[0]
            dev = x2.device
            xn = torch.empty((rows, D), dtype=cdt, device=dev)
            mean = torch.empty(rows, dtype=torch.float32, device=dev)
            rstd = torch.empty(rows, dtype=torch.float32, device=dev)
            u = torch.empty((rows, M), dtype=cdt, device=dev)
            h = torch.empty((rows, M), dtype=cdt, device=dev)
            y = torch.empty((rows, D), dtype=cdt, device=dev)
            check(lib.vitmi_ff_block_fwd(_dt(cdt), rows, D, M, _p(x2), _p(g), _p(ln_b.detach().float()), eps,
                                         _p(compute_copy(w1, cdt)), _p(b1.detach().float()),
                                         _p(compute_copy(w2, cdt)), _p(b2.detach().float()), _p(xn), _p(mean),
                                         _p(rstd), _p(u), _p(h), _p(y), _stream()), "vitmi_ff_block_fwd")
        else:
            if ln_w is not None:
                g = ln_w.detach().float()
                xn, mean, rstd = layernorm_fwd(x2, g, ln_b.detach().float(), eps)
            else:
                g = mean = rstd = None
                xn = x2
            # u = gelu'(pre-activation), evaluated in the forward epilogue beside gelu itself (same erf and exp)
            h, u = linear_fwd(xn, compute_copy(w1, cdt), b1.detach().float(), epi=EPI_GELU_D, want_aux=True)
            if drop is None:
                y, _ = linear_fwd(h, compute_copy(w2, cdt), b2.detach().float(), res=x2 if add_residual else None)
            else:
                dropout(h, drop[0], drop[1], out=h)            # the saved h is the dropped one: dW2 = dy^T h
                y, _ = linear_fwd(h, compute_copy(w2, cdt), b2.detach().float())
                dropout(y, drop[0], drop[2], out=y)
                if add_residual:
                    y = add(y, x2.contiguous())
        ctx.whole = whole
        ctx.drop = drop
        # what `u` holds: gelu'(pre-activation) (multiplied in by the backward epilogue), or, when the whole-block call ran
        # with the "erf" pair, the pre-activation itself (the backward epilogue evaluates the derivative)
        ctx.aux_epi = EPI_DGELU if (whole and lib.vitmi_get_gelu_mode()) else EPI_MULAUX
        ctx.save_for_backward(x2, g, mean, rstd, xn if ln_w is not None else None, u, h, w1, w2)
        ctx.meta = (shape, add_residual, cdt, ln_w is not None, ln_w.dtype if ln_w is not None else None, x.dtype)
        ctx.vec_params = (b1, ln_w, ln_b, b2)            # (db1, dgamma, dbeta, db2) slots of the gathered vector sums
        return y.view(*shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd, xn, u, h, w1, w2 = ctx.saved_tensors
        shape, add_residual, cdt, has_ln, ln_dtype, in_dtype = ctx.meta
        rp = _PS2
        if rp is not None:
            dy2, du, s_b1, s_ln = _kept(ctx)
            seg = x2.shape[0] // rp.coef.numel()
            xn_ = xn if xn is not None else x2
            dys = seg_scale_rows(dy2, rp.coef, seg)
            xs = seg_scale_rows(xn_, rp.coef, seg)
            dw2, dw1 = linear_bwd_w_pair(dys, h, du, xs)
            db2 = colsum(dys).to(w2.dtype)
            db1 = seg_combine(s_b1, rp.coef).to(w1.dtype)
            dg = db = None
            if has_ln:
                db, dg = seg_combine(s_ln[0], rp.coef).to(ln_dtype), seg_combine(s_ln[1], rp.coef).to(ln_dtype)
            return (_no_grad_like(shape, in_dtype, dy.device), dg, db, dw1.to(w1.dtype), db1, dw2.to(w2.dtype), db2, None,
                    None, None, None)
        dy2 = _as2d(dy.contiguous())
        if dy2.dtype != cdt:
            dy2 = cast(dy2, cdt)
        ps = _PS
        drop = ctx.drop
        if drop is not None:
            if ps is not None:
                raise NotImplementedError("per-sample norms through a block with dropout")
            if xn is None:
                xn = x2
            dyd = dropout(dy2, drop[0], drop[2])
            dw2 = linear_bwd_w(dyd, h).to(w2.dtype)
            db2 = colsum(dyd).to(w2.dtype)
            du = linear_bwd_x(dyd, compute_copy(w2, cdt), epi=ctx.aux_epi, aux=u, w_t=weight_t(w2, cdt))
            dropout(du, drop[0], drop[1], out=du)
            dw1 = linear_bwd_w(du, xn).to(w1.dtype)
            db1 = colsum(du).to(w1.dtype)
            dxn = linear_bwd_x(du, compute_copy(w1, cdt), w_t=weight_t(w1, cdt))
            dg = db = None
            if has_ln:
                dx, dg, db = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2 if add_residual else None)
                dg, db = dg.to(ln_dtype), db.to(ln_dtype)
            else:
                dx = add(dxn, dy2) if add_residual else dxn
            dx = dx.view(shape)
            if dx.dtype != in_dtype:
                dx = cast(dx, in_dtype)
            return dx, dg, db, dw1, db1, dw2, db2, None, None, None, None
        if ps is not None:
            xn_ = xn if xn is not None else x2
            ps_weight(ps, dy2, h)
            ps_bias(ps, dy2)
            du = linear_bwd_x(dy2, compute_copy(w2, cdt), epi=ctx.aux_epi, aux=u, w_t=weight_t(w2, cdt))
            ps_weight(ps, du, xn_)
            s_b1 = ps_bias(ps, du, want_sums=ps.keep)
            dxn = linear_bwd_x(du, compute_copy(w1, cdt), w_t=weight_t(w1, cdt))
            s_ln = None
            if has_ln:
                s_ln = ps_layernorm(ps, dxn, x2, mean, rstd, want_sums=ps.keep)
                dx, _, _ = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2 if add_residual else None, need_affine=False)
            else:
                dx = add(dxn, dy2) if add_residual else dxn
            if ps.keep:
                ctx.kept = (dy2, du, s_b1, s_ln)
            dx = dx.view(shape)
            if dx.dtype != in_dtype:
                dx = cast(dx, in_dtype)
            return (dx,) + (None,) * 10
        if ctx.whole and dy2.is_contiguous():
            rows, D = x2.shape
            M = w1.shape[0]
            dev, f32 = x2.device, torch.float32
            dx = torch.empty((rows, D), dtype=cdt, device=dev)
            dg, db = torch.empty(D, dtype=f32, device=dev), torch.empty(D, dtype=f32, device=dev)
            dw1, db1 = torch.empty((M, D), dtype=f32, device=dev), torch.empty(M, dtype=f32, device=dev)
            dw2, db2 = torch.empty((D, M), dtype=f32, device=dev), torch.empty(D, dtype=f32, device=dev)
            key = ("ff", _dt(cdt), rows, D, M)
            nbytes = _block_ws_bytes.get(key)
            if nbytes is None:
                nbytes = _block_ws_bytes[key] = int(lib.vitmi_ff_block_bwd_workspace_bytes(_dt(cdt), rows, D, M))
            ws = _ws(nbytes, dev)
            defer = D % 8 == 0 and M % 8 == 0 and _defer_ok(cdt, rows, D, (w1, w2))
            strips = 0
            if (defer and ln_dtype == f32 and w1.dtype == f32 and w2.dtype == f32
                    and _defer_sums_ok(ctx.vec_params)):
                strips = int(lib.vitmi_ff_block_bwd_sums_strips(_dt(cdt), rows, D, M))
            late = strips > 0                # the vector sums ride on the queue as well
            check(lib.vitmi_ff_block_bwd(_dt(cdt), rows, D, M, _p(dy2), _p(x2), _p(xn), _p(u), _p(h), _p(g), _p(mean),
                                         _p(rstd), _p(compute_copy(w1, cdt)), _p(weight_t(w1, cdt)),
                                         _p(compute_copy(w2, cdt)), _p(weight_t(w2, cdt)), _p(dx),
                                         None if late else _p(dg), None if late else _p(db),
                                         None if defer else _p(dw1), None if late else _p(db1),
                                         None if defer else _p(dw2), None if late else _p(db2),
                                         _p(ws), _stream()), "vitmi_ff_block_bwd")
            if defer:
                du = ws.data_ptr() + lib.vitmi_ff_block_bwd_du_offset(_dt(cdt), rows, D, M)
                _defer_jobs(rows, dev, [(_p(dy2), D, _p(h), M, dw2, D, M, w2), (du, M, _p(xn), D, dw1, M, D, w1)],
                            (dy2, h, xn, ws))
            if late:
                part = ws.data_ptr() + lib.vitmi_ff_block_bwd_sums_offset(_dt(cdt), rows, D, M)
                _defer_sum_job(dev, part, strips, M, D, (db1, dg, db, db2), ctx.vec_params, (ws,))
            dx = dx.view(shape)
            if dx.dtype != in_dtype:
                dx = cast(dx, in_dtype)
            return (dx, dg.to(ln_dtype), db.to(ln_dtype), dw1.to(w1.dtype), db1.to(w1.dtype), dw2.to(w2.dtype),
                    db2.to(w2.dtype), None, None, None, None)
        if xn is None:
            xn = x2
        dw2 = linear_bwd_w(dy2, h).to(w2.dtype)
        fuse_db2 = has_ln and add_residual
        db2 = None if fuse_db2 else colsum(dy2).to(w2.dtype)
        du, db1 = linear_bwd_x(dy2, compute_copy(w2, cdt), epi=ctx.aux_epi, aux=u, w_t=weight_t(w2, cdt),
                               want_colsum=True)
        dw1 = linear_bwd_w(du, xn).to(w1.dtype)
        db1 = db1.to(w1.dtype)
        dxn = linear_bwd_x(du, compute_copy(w1, cdt), w_t=weight_t(w1, cdt))
        dg = db = None
        if has_ln:
            if fuse_db2:
                dx, dg, db, rs = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2, want_dres_colsum=True)
                db2 = rs.to(w2.dtype)
            else:
                dx, dg, db = layernorm_bwd(dxn, x2, g, mean, rstd, dres=dy2 if add_residual else None)
            dg, db = dg.to(ln_dtype), db.to(ln_dtype)
        else:
            dx = add(dxn, dy2) if add_residual else dxn
        dx = dx.view(shape)
        if dx.dtype != in_dtype:
            dx = cast(dx, in_dtype)
        return dx, dg, db, dw1, db1, dw2, db2, None, None, None, None


class MeanPoolFn(torch.autograd.Function):
    """x.mean(dim=1) for pool == 'mean' (mymodel/ViT.py:164)."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = x.contiguous()
        B, N, D = x.shape
        out = torch.empty((B, D), dtype=x.dtype, device=x.device)
        check(lib.vitmi_meanpool_fwd(_p(x), _p(out), _dt(x), B, N, D, _stream()), "vitmi_meanpool_fwd")
        ctx.n = N
        return out

    @staticmethod
    def backward(ctx, dout):
        if _PS2 is not None:                # a replay forms no input gradients
            return _no_grad_like((dout.shape[0], ctx.n, dout.shape[1]), dout.dtype, dout.device)
        dout = dout.contiguous()
        B, D = dout.shape
        dx = torch.empty((B, ctx.n, D), dtype=dout.dtype, device=dout.device)
        check(lib.vitmi_meanpool_bwd(_p(dout), _p(dx), _dt(dout), B, ctx.n, D, _stream()), "vitmi_meanpool_bwd")
        return dx
