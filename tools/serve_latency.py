#!/usr/bin/env python3
"""Forward latency at serving batch sizes, eager against GraphedInference (one HIP graph per shape), bf16, no autograd
graph.  usage: serve_latency.py [vit_b16 | cifar]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT  # noqa: E402
from visiontransformer_amd.graphs import GraphedInference  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
if which == "cifar":
    kw, size = dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=12, heads=12, mlp_dim=512), 32
else:
    kw, size = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072,
                    dim_head=64), 224
torch.manual_seed(3)
model = ViT.VIT(**kw).to("cuda:0").eval()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (1, 8, 32, 256):
    x = torch.randn(B, 3, size, size, device="cuda:0")
    with torch.no_grad():
        eager = timed(lambda: model(x))
    gi = GraphedInference(model, x)
    graph = timed(lambda: gi(x))
    print(f"{which} batch {B:3d}: eager {eager:.3f} ms ({B / eager * 1e3:.0f} images/s), graph {graph:.3f} ms "
          f"({B / graph * 1e3:.0f} images/s)", flush=True)
