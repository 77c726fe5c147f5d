#!/usr/bin/env python3
"""Which host operations of one cifar_mask training step launch the fill and copy kernels (torch.profiler, eager step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT_mask  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

dev = "cuda:0"
kw = dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=12, heads=12, mlp_dim=512)
m = ViT_mask.VIT(**kw).to(dev).set_compute_dtype(torch.bfloat16).train()
m.train_method = "mask"
opt = FusedAdam(m.parameters(), lr=1e-3)
x = torch.randn(256, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (256,), device=dev)
um = torch.zeros(1, 65, dtype=torch.long, device=dev)
um[0, 3:40:2] = 1


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(m(x, unk_mask=um), y)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    for k in e.kernels:
        if "Fill" in k.name or "copy" in k.name.lower() or "elementwise" in k.name:
            chain, p = [], e
            while p is not None and len(chain) < 5:
                chain.append(p.name)
                p = p.cpu_parent
            key = (k.name[:50], " <- ".join(chain))
            seen[key] = seen.get(key, 0) + 1
for (k, c), n in sorted(seen.items(), key=lambda t: -t[1]):
    print(n, k, "|", c)
