#!/usr/bin/env python3
"""Streaming attention kernels against the one-workgroup-per-pair kernels on one box, one process: outputs must be
bit-identical; the two are timed alternately.  usage: attn_stream_cmp.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


B, H, DH = 256, 12, 64
for N in [int(a) for a in sys.argv[1:]] or [197, 145, 160, 176, 192, 208]:
    torch.manual_seed(N)
    qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
    res = {}
    for on in (0, 1):
        lib.vitmi_debug_set_attn_stream(on)
        o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
        dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)
        torch.cuda.synchronize()
        res[on] = (o.clone(), lse.clone(), dqkv.clone())
    same = [torch.equal(a, b) for a, b in zip(res[0], res[1])]
    t = {0: [[], []], 1: [[], []]}
    for rep in range(3):
        for on in (0, 1):
            lib.vitmi_debug_set_attn_stream(on)
            t[on][0].append(timed(lambda: ops.attn_fwd(qkv, B, N, H, DH, False)))
            t[on][1].append(timed(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)))
    print(f"N {N}: identical out/lse/dqkv {same}; fwd per-pair {min(t[0][0]):.1f} us, streaming {min(t[1][0]):.1f} us; "
          f"bwd per-pair {min(t[0][1]):.1f} us, streaming {min(t[1][1]):.1f} us", flush=True)
lib.vitmi_debug_set_attn_stream(1)
