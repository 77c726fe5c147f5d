#!/usr/bin/env python3
"""torch's own DistributedDataParallel around the drop-in module, as trainer.py:184 wraps it
(find_unused_parameters=True): two steps on one rank with the RCCL backend.  Launch with torch.distributed.run."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT_mask  # noqa: E402

rank = int(os.environ["RANK"])
local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
torch.manual_seed(0)
model = ViT_mask.VIT(32, 4, 96, 10, 2, 6, 128).to(dev)
model.train_method = "mask"
model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                  find_unused_parameters=True)
opt = torch.optim.Adam(model.parameters(), lr=2e-3)
x = torch.randn(32, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (32,), device=dev)
unk = torch.zeros(1, 65, dtype=torch.long, device=dev)
unk[0, 3:30] = 1
losses = []
for step in range(6):
    opt.zero_grad()
    # without a mask the unknown row of the positional table takes no gradient: the unused-parameter path
    loss = torch.nn.functional.cross_entropy(model(x, unk_mask=unk if step % 2 else None), y)
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
print("losses", " ".join(f"{v:.4f}" for v in losses), flush=True)
assert losses[-1] < losses[0]
dist.barrier()
dist.destroy_process_group()
print("ddp wrap ok", flush=True)
