#!/usr/bin/env python3
"""Is the forward attention kernel's time set by where its operands lie?  The same 3072 (image, head) pairs of 197 x 64
run as B=256, H=12 (a head's rows are 128-byte pieces 4608 bytes apart) and as B=3072, H=1 (384 bytes apart: every
third 128-byte line of one contiguous block)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402


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


N, DH = 197, 64
for B, H in ((256, 12), (3072, 1)):
    qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
    o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
    tf = min(timed(lambda: ops.attn_fwd(qkv, B, N, H, DH, False)) for _ in range(3))
    tb = min(timed(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)) for _ in range(3))
    print(f"B {B} H {H}: fwd {tf:.1f} us, bwd {tb:.1f} us", flush=True)
