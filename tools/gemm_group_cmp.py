#!/usr/bin/env python3
"""Tile-walk grouping of the persistent nt kernel (vitmi_debug_set_nt_group) on the encoder shapes, every epilogue:
microseconds per launch for gm = 0 (n fastest) and grouped walks, results checked against gm = 0 bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, EPI_DGELU, EPI_GELU, EPI_GELU_D, EPI_MULAUX, lib  # noqa: E402

T = 256 * 197
bf = torch.bfloat16
GMS = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,4,8,16").split(",")]
if len(sys.argv) > 2:
    lib.vitmi_debug_set_nt_tail_split(int(sys.argv[2]))


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


for (N, K) in [(3072, 768), (2304, 768), (768, 3072), (768, 2304), (768, 768)]:
    x = torch.randn(T, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(T, N, device="cuda").to(bf)
    aux = torch.randn(T, N, device="cuda").to(bf)
    y = torch.empty(T, N, device="cuda", dtype=bf)
    cases = {
        "plain": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16),
        "bias+res": lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, bias=bias, res=res, ldr=N),
    }
    if N == 3072:
        cases["gelu+aux"] = lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, epi=EPI_GELU, bias=bias,
                                                 aux=aux, ldaux=N)
        cases["gelu_d"] = lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, epi=EPI_GELU_D, bias=bias,
                                               aux=aux, ldaux=N)
        cases["mulaux"] = lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, epi=EPI_MULAUX, aux=aux,
                                               ldaux=N)
        cases["dgelu"] = lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16, epi=EPI_DGELU, aux=aux,
                                              ldaux=N)
    for name, fn in cases.items():
        line = f"N={N} K={K} {name:9s}:"
        ref = None
        for gm in GMS:
            lib.vitmi_debug_set_nt_group(gm)
            fn()
            out = y.clone()
            if ref is None:
                ref = out
            ok = torch.equal(out, ref) or name in ("gelu+aux", "gelu_d")
            us = timed(fn) * 1e3
            line += f"  gm{gm} {us:6.1f}us{'' if ok else ' MISMATCH'}"
        print(line, flush=True)
lib.vitmi_debug_set_nt_group(-1)
