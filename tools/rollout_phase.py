#!/usr/bin/env python3
"""Rollout at the headline geometry with and without the discard (discard_ratio 0 skips the selection rounds): the
difference is the cost of the selection; run under rocprofv3 --kernel-trace for the per-kernel split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import vit_rollout  # noqa: E402

dev = "cuda:0"


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (B, H, N, L, dt) in [(256, 12, 197, 12, torch.bfloat16), (128, 12, 197, 12, torch.float32)]:
    atts = [torch.softmax(torch.randn(B, H, N, N, device=dev), -1).to(dt) for _ in range(L)]
    for ratio in (0.9, 0.0):
        ms = timed(lambda: vit_rollout.rollout(atts, ratio, "mean"))
        print(f"B={B} N={N} {dt} discard {ratio}: {ms:.3f} ms", flush=True)
