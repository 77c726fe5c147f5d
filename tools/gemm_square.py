#!/usr/bin/env python3
"""nt GEMM rate on square problems (4096^3, 8192^3) per tile configuration, random operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, lib  # noqa: E402

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


for S in (4096, 8192):
    x = (torch.rand(S, S, device="cuda") * 2 - 1).to(bf)
    w = (torch.rand(S, S, device="cuda") * 2 - 1).to(bf)
    y = torch.empty(S, S, device="cuda", dtype=bf)
    fl = 2.0 * S ** 3
    for cfg in cfgs:
        lib.vitmi_debug_set_nt_config(cfg)
        ms = timeit(lambda: ops.gemm_raw(BF16, S, S, S, x, S, 1, w, S, 1, y, S, BF16))
        print(f"nt {S}^3 cfg {cfg}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    ms = timeit(lambda: torch.matmul(x, w.t()))
    print(f"nt {S}^3 torch.matmul: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
lib.vitmi_debug_set_nt_config(-1)
