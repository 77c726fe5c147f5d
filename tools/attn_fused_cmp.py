#!/usr/bin/env python3
"""bf16 attention backward for heads of 64: the one-kernel variant against the dQ + dK/dV pair, one process, one box.
Both are compared with an fp32 autograd reference on the same bf16 inputs and timed alternately."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def reference(qkv, do, B, N, H, DH):
    x = qkv.float().requires_grad_(True)
    q, k, v = x.view(B, N, 3, H, DH).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) / DH ** 0.5, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, H * DH)
    o.backward(do.float())
    return x.grad


shapes = [tuple(int(v) for v in sys.argv[1:5])] if len(sys.argv) >= 5 else [(256, 197, 12, 64), (64, 145, 12, 64), (32, 208, 12, 64), (300, 193, 3, 64), (7, 200, 5, 64)]
for B, N, H, DH in shapes:
    torch.manual_seed(N + DH)
    qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
    o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
    nb = min(B, 16)
    ref = reference(qkv[:nb], do[:nb], nb, N, H, DH)
    res = {}
    modes = (0, 1, 2)           # two kernels; one kernel; one kernel, operands fetched a pair ahead (193..208 tokens)
    for on in modes:
        lib.vitmi_debug_set_attn_bwd_fused(on)
        res[on] = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH).clone()
        again = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)
        torch.cuda.synchronize()
        assert torch.equal(res[on], again), f"mode {on}: two calls differ"
    err = {on: ((res[on][:nb].float() - ref).abs().max() / ref.abs().max()).item() for on in modes}
    errl = {on: ((res[on][-nb:].float() - reference(qkv[-nb:], do[-nb:], nb, N, H, DH)).abs().max() / ref.abs().max()).item() for on in modes}
    same12 = torch.equal(res[1], res[2])
    finite = all(bool(torch.isfinite(res[on].float()).all()) for on in modes)
    t = {on: [] for on in modes}
    for rep in range(3):
        for on in modes:
            lib.vitmi_debug_set_attn_bwd_fused(on)
            t[on].append(timed(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)))
    print(f"B {B} N {N} H {H} dh {DH}: error against fp32 (first / last images) " +
          ", ".join(f"mode {on}: {err[on]:.2e} / {errl[on]:.2e}" for on in modes) +
          f"; modes 1 and 2 identical {same12}; finite {finite}; " + ", ".join(f"mode {on}: {min(t[on]):.1f} us" for on in modes), flush=True)
lib.vitmi_debug_set_attn_bwd_fused(0)
