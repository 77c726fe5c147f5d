#!/usr/bin/env python3
"""Times the attention kernels on the ViT-B/16 shape (B=256, N=197, H=12, dh=64) and the cifar
shape (B=256, N=65, H=12, dh=8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for B, N, H, DH in [(256, 197, 12, 64), (256, 65, 12, 8)]:
    qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
    o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
    fl = 4.0 * B * H * N * N * DH
    ms = timeit(lambda: ops.attn_fwd(qkv, B, N, H, DH, False))
    print(f"attn fwd        B{B} N{N} H{H} dh{DH}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
    ms = timeit(lambda: ops.attn_fwd(qkv, B, N, H, DH, True))
    print(f"attn fwd+export B{B} N{N} H{H} dh{DH}: {ms:7.3f} ms  export {B * H * N * N * 2 / ms / 1e6:7.1f} GB/s")
    ms = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, DH))
    print(f"attn bwd        B{B} N{N} H{H} dh{DH}: {ms:7.3f} ms {2.5 * fl / ms / 1e9:7.1f} TF")
