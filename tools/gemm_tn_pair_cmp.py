#!/usr/bin/env python3
"""The two weight gradients of a sub-block at the benchmark's size: one batched launch against two launches, alternating
in one process on one box."""
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


T = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
for name, (Pa, Qa, Pb, Qb) in {"mlp": (768, 3072, 3072, 768), "attention": (768, 768, 2304, 768)}.items():
    g = torch.Generator(device="cuda").manual_seed(3)
    ts = [torch.randn(T, c, device="cuda", generator=g).to(torch.bfloat16) for c in (Pa, Qa, Pb, Qb)]
    res = {0: [], 1: []}
    for rep in range(3):
        for on in (1, 0):
            lib.vitmi_debug_set_tn_batch(on)
            res[on].append(timed(lambda: ops.linear_bwd_w_pair(*ts)))
    lib.vitmi_debug_set_tn_batch(1)
    fl = 2.0 * T * (Pa * Qa + Pb * Qb)
    print(f"{name}: one launch {min(res[1]):.1f} us ({fl / min(res[1]) / 1e6:.0f} TFLOP/s), two launches {min(res[0]):.1f} us "
          f"({fl / min(res[0]) / 1e6:.0f} TFLOP/s)", flush=True)
