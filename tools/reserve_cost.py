#!/usr/bin/env python3
"""What leaving compute units to a collective costs the ViT-B/16 training step on one GPU: vitmi_set_reserved_cus(k)
for k in 0, 8, 16, 32, alternating in one process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT  # noqa: E402
from visiontransformer_amd._lib import check, lib  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

KW = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072, dim_head=64)
dev = torch.device("cuda:0")
torch.manual_seed(1001)
model = ViT.VIT(**KW).to(dev).train()
opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)
x = torch.randn(256, 3, 224, 224, device=dev)
y = torch.randint(0, 100, (256,), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(model(x), y).backward()
    opt.step()


for _ in range(5):
    step()
ks = (0, 8, 16, 32)
res = {k: [] for k in ks}
for rep in range(3):
    for k in ks:
        check(lib.vitmi_set_reserved_cus(k), "vitmi_set_reserved_cus")
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 10 * 1e3)
check(lib.vitmi_set_reserved_cus(0), "vitmi_set_reserved_cus")
for k in ks:
    print(f"reserved {k:3d}: {min(res[k]):.3f} ms/step ({min(res[k]) / min(res[0]) - 1:+.1%})", flush=True)
