#!/usr/bin/env python3
"""cProfile of the host side of a training step at the launch-bound cifar geometry (where the step time is the
Python/ctypes time, not kernel time): shows which host calls to trim."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import ViT_mask  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = ViT_mask.VIT(32, 4, 192, 10, 12, 12, 512).to(dev)
m.train_method = "mask"
m.train()
opt = FusedAdam(m.parameters(), lr=1e-3)
x = torch.randn(256, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (256,), device=dev)
unk = torch.zeros(1, 65, dtype=torch.long, device=dev)
unk[0, 5:30] = 1


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(m(x, unk_mask=unk), y)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
