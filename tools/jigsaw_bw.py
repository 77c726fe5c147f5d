#!/usr/bin/env python3
"""HBM rate of the jigsaw patch gather (vitmi_jigsaw_gather): bytes read + written / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import ops  # noqa: E402

dev = "cuda:0"
for (B, C, img, p, dt) in [(256, 3, 224, 16, torch.float32), (256, 3, 224, 16, torch.bfloat16),
                           (256, 3, 32, 4, torch.float32), (1024, 3, 224, 16, torch.float32)]:
    L = (img // p) ** 2
    x = torch.randn(B, C, img, img, device=dev).to(dt)
    src = torch.arange(L, dtype=torch.int32)
    perm = torch.randperm(L // 2)
    src[: L // 2] = perm.to(torch.int32)
    src = src.to(dev)
    for _ in range(3):
        y = ops.jigsaw_gather(x, p, src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y = ops.jigsaw_gather(x, p, src)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    nbytes = 2 * x.numel() * x.element_size()
    print(f"B={B} C={C} {img}px p={p} {dt}: {us:.1f} us/launch, {nbytes / us / 1e3:.0f} GB/s", flush=True)
