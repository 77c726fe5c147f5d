#!/usr/bin/env python3
"""Yardstick only: what the vendor BLAS behind torch.matmul reaches on the ViT-B/16 GEMM shapes
(nt = activations x W^T, tn = weight gradient).  Not used by the library."""
import torch

T = 256 * 197
bf = torch.bfloat16


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


for N, K in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn(T, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
    dy = torch.randn(T, N, device="cuda").to(bf)
    fl = 2.0 * T * N * K
    ms = timeit(lambda: torch.matmul(x, w.t()))
    print(f"nt {T}x{N}x{K}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    ms = timeit(lambda: torch.matmul(dy.t(), x))
    print(f"tn {N}x{K}x{T}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    ms = timeit(lambda: torch.matmul(dy, w))
    print(f"nn {T}x{K}x{N}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
