#!/usr/bin/env python3
"""Times the GEMM entry points on the ViT-B/16 (batch 256) shapes with random operands.
Runs on the GPU box: python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402

DEV = "cuda:0"
T = 256 * 197


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


def main():
    from visiontransformer_amd._lib import lib
    bf = torch.bfloat16
    for cfg in (0, 1, -1):
        lib.vitmi_debug_set_tn_config(cfg)
        print(f"--- tn tile config {cfg} (0=128x128 1=256x256 -1=auto)")
        run(bf, dw=True)


def run(bf, dw):
    rows = []
    for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
        x = torch.randn(T, K, device=DEV).to(bf)
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(bf)
        b = torch.randn(N, device=DEV)
        r = torch.randn(T, N, device=DEV).to(bf)
        dy = torch.randn(T, N, device=DEV).to(bf)
        wt = w.t().contiguous()
        fl = 2.0 * T * N * K
        ms = timeit(lambda: ops.linear_fwd(x, w, b))
        rows.append((f"{name} fwd nt  [{T}x{N}x{K}]", ms, fl / ms / 1e9))
        if name == "fc1":
            ms = timeit(lambda: ops.linear_fwd(x, w, b, epi=ops.EPI_GELU, want_aux=True))
            rows.append((f"{name} fwd nt+gelu", ms, fl / ms / 1e9))
        if name in ("out", "fc2"):
            ms = timeit(lambda: ops.linear_fwd(x, w, b, res=r))
            rows.append((f"{name} fwd nt+res", ms, fl / ms / 1e9))
        ms = timeit(lambda: ops.linear_bwd_x(dy, w, w_t=wt))
        rows.append((f"{name} dx  nt(w^T)", ms, fl / ms / 1e9))
        if dw:
            ms = timeit(lambda: ops.linear_bwd_w(dy, x))
            rows.append((f"{name} dw  tn", ms, fl / ms / 1e9))
    for n, ms, tf in rows:
        print(f"{n:36s} {ms:8.3f} ms  {tf:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
