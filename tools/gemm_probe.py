#!/usr/bin/env python3
"""Where does the nt/tn GEMM time go?  Same launch with (a) real operands, (b) every row aliased
onto row 0 (lda = ldb = 0: all loads hit the caches), on the ViT-B/16 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, F32, lib, check  # noqa: E402

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
    y = torch.empty(T, N, device="cuda", dtype=bf)
    fl = 2.0 * T * N * K
    for name, lda, ldb in (("real", K, K), ("A aliased", 0, K), ("B aliased", K, 0), ("both aliased", 0, 0)):
        ms = timeit(lambda: ops.gemm_raw(BF16, T, N, K, x, lda, 1, w, ldb, 1, y, N, BF16))
        print(f"nt {T}x{N}x{K} {name:14s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
    dy = torch.randn(T, N, device="cuda").to(bf)
    dw = torch.empty(N, K, device="cuda", dtype=torch.float32)
    ws = torch.empty(max(lib.vitmi_gemm_wgrad_workspace_bytes(BF16, N, K, T), 16), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for name, ldu, ldv in (("real", N, K), ("both aliased", 0, 0)):
        ms = timeit(lambda: check(lib.vitmi_gemm_wgrad(BF16, N, K, T, dy.data_ptr(), max(ldu, 0), x.data_ptr(),
                                                      max(ldv, 0), dw.data_ptr(), K, ws.data_ptr(), s)))
        print(f"tn {N}x{K}x{T} {name:14s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
