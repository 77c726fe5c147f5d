#!/usr/bin/env python3
"""Compare nt tile configurations (vitmi_debug_set_nt_config) on the ViT-B/16 shapes: checks each
configuration against a torch matmul and prints its rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, lib  # noqa: E402

T = 256 * 197
bf = torch.bfloat16
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,3").split(",")]


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


for N, K in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]:
    x = torch.randn(T, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
    ref = (x[:4096].float() @ w.float().t())
    fl = 2.0 * T * N * K
    for cfg in cfgs:
        lib.vitmi_debug_set_nt_config(cfg)
        y = torch.zeros(T, N, device="cuda", dtype=bf)
        ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16)
        err = ((y[:4096].float() - ref).abs().max() / ref.abs().max()).item()
        tail = (y[-300:].float() - x[-300:].float() @ w.float().t()).abs().max().item()
        ms = timeit(lambda: ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16))
        ms_al = timeit(lambda: ops.gemm_raw(BF16, T, N, K, x, 0, 1, w, 0, 1, y, N, BF16))
        xz, wz = torch.zeros_like(x), torch.zeros_like(w)
        ms_z = timeit(lambda: ops.gemm_raw(BF16, T, N, K, xz, K, 1, wz, K, 1, y, N, BF16))
        print(f"nt {T}x{N}x{K} cfg {cfg}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF  rel err {err:.2e} tail {tail:.2e}"
              f" | aliased rows {fl / ms_al / 1e9:7.1f} TF | zero operands {fl / ms_z / 1e9:7.1f} TF", flush=True)
lib.vitmi_debug_set_nt_config(-1)
