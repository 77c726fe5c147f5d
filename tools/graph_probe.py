#!/usr/bin/env python3
"""cifar-geometry masked training step with and without the HIP graph of forward + backward."""
import os, random, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT_mask, ops
from visiontransformer_amd.graphs import GraphedForwardBackward
from visiontransformer_amd.masking_generator import MaskingGenerator
from visiontransformer_amd.optim import FusedAdam

dev = torch.device("cuda", 0)
kw = dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=12, heads=12, mlp_dim=512)
gen = MaskingGenerator((8, 8), 32, min_num_patches=4)
x = torch.randn(256, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (256,), device=dev)
ce = torch.nn.functional.cross_entropy


def mask(i):
    random.seed(1234 + i), np.random.seed(1234 + i)
    return ops.upload(MaskingGenerator.unk_mask_from_flat(gen.masked_indexes()[0]).astype(np.int64), dev)


def make():
    torch.manual_seed(1001)
    m = ViT_mask.VIT(**kw).to(dev).train()
    m.train_method = "mask"
    return m, FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)


def timed(step, n=60, warm=10):
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, warm + n):
        loss = step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(loss)


m1, o1 = make()
def eager(i):
    o1.zero_grad()
    loss = ce(m1(x, unk_mask=mask(i)), y)
    loss.backward()
    o1.step()
    return loss
ms1, l1 = timed(eager)
m2, o2 = make()
g = GraphedForwardBackward(m2, ce, x, y, mask(0))
def graphed(i):
    loss = g(x, y, mask(i))
    o2.step()
    return loss
ms2, l2 = timed(graphed)
print(f"eager {ms1:.3f} ms/step loss {l1:.6f} | graphed {ms2:.3f} ms/step loss {l2:.6f} | {256 / ms2 * 1e3:.0f} img/s")
same = all(torch.equal(a, b) for a, b in zip(m1.parameters(), m2.parameters()))
print("parameters identical after the same steps:", same)
