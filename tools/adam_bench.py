#!/usr/bin/env python3
"""Times the multi-tensor Adam step on the ViT-B/16 parameter set (85.8 M parameters, 30 bytes moved per parameter)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

m = ViT.VIT(224, 16, 768, 100, 12, 12, 3072, dim_head=64).to("cuda")
opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)
x = torch.randn(4, 3, 224, 224, device="cuda")
torch.nn.functional.cross_entropy(m(x), torch.zeros(4, dtype=torch.long, device="cuda")).backward()
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    opt.step()
b.record()
torch.cuda.synchronize()
n = sum(p.numel() for p in m.parameters())
us = a.elapsed_time(b) / 20 * 1e3
print(f"adam + transposed copies: {us:.1f} us per step, {n / 1e6:.1f} M parameters")
