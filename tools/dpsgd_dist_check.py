#!/usr/bin/env python3
"""Batched DP-SGD through the gradient reducer on RCCL (launched by torch.distributed.run, any number of ranks, one
GPU each): the step with `reducer=` must leave the same gradient as the step without on one rank, and identical
parameters on every rank after a few steps with differing data."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

from visiontransformer_amd import ViT, dpsgd  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402
from visiontransformer_amd.parallel import GradReducer, broadcast_parameters  # noqa: E402

kw = dict(image_size=32, patch_size=4, embed_dim=64, num_classes=10, depth=2, heads=4, mlp_dim=128)
torch.manual_seed(5)
model = ViT.VIT(**kw).to(dev).train()
broadcast_parameters(model, src=0)
torch.manual_seed(100 + rank)
x = torch.randn(8, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (8,), device=dev)

gen = torch.Generator(device=dev).manual_seed(9)
dpsgd.clipped_noised_gradients(model, x, y, 0.5, 1.0, generator=gen)
plain = [p.grad.clone() for p in model.parameters()]
reducer = GradReducer(model, bucket_mb=1, average=False)
gen = torch.Generator(device=dev).manual_seed(9)
dpsgd.clipped_noised_gradients(model, x, y, 0.5, 1.0, generator=gen, reducer=reducer)
if world == 1:
    worst = max(float((p.grad - g).abs().max()) for p, g in zip(model.parameters(), plain))
    print(f"max gradient difference with / without the reducer on one rank: {worst:.3e}")
    assert worst == 0.0
opt = FusedAdam(model.parameters(), lr=1e-3)
for step in range(3):
    loss, norms = dpsgd.clipped_noised_gradients(model, x, y, 0.5, 1.0, generator=gen, reducer=reducer)
    opt.step()
flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
ref = flat.clone()
dist.broadcast(ref, src=0)
print(f"rank {rank}: max parameter difference to rank 0: {float((flat - ref).abs().max()):.3e}")
assert torch.equal(flat, ref)
dist.barrier()
dist.destroy_process_group()
print("dpsgd dist ok")
