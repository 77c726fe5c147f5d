#!/usr/bin/env python3
"""Cost of the nt GEMM epilogues on the encoder shapes: plain, +bias, +bias+residual, GELU (+aux), dGELU (+aux)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, EPI_DGELU, EPI_GELU, EPI_NONE  # noqa: E402

T = 256 * 197
bf = torch.bfloat16


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
    return a.elapsed_time(b) / n


for (N, K) in [(768, 3072), (768, 768), (3072, 768)]:
    x = torch.randn(T, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(T, N, device="cuda").to(bf)
    aux = torch.randn(T, N, device="cuda").to(bf)
    y = torch.empty(T, N, device="cuda", dtype=bf)
    fl = 2.0 * T * N * K
    cases = {
        "plain": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16),
        "bias": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, bias=bias),
        "bias+res": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, bias=bias, res=res, ldr=N),
        "gelu+aux": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, epi=EPI_GELU, bias=bias, aux=aux,
                                         ldaux=N),
        "dgelu": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, epi=EPI_DGELU, aux=aux, ldaux=N),
    }
    line = f"N={N} K={K}:"
    for name, fn in cases.items():
        ms = timed(fn)
        line += f"  {name} {ms * 1e3:.0f} us ({fl / ms / 1e9:.0f} TF)"
    print(line, flush=True)
