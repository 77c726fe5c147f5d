#!/usr/bin/env python3
"""Rehearsal of the data-parallel step of the HIP modules: every stage logs a line per rank.  Launch with
torch.distributed.run.  Backend "nccl" (RCCL; one rank per GPU) runs the collectives on device buffers; with
"gloo" several ranks can share ONE GPU: broadcast_parameters and GradReducer then move the buckets through host
copies (gloo's own device-tensor collectives stalled at varying places with two processes on one ROCm device, so
nothing here hands gloo a device tensor).  REHEARSE_BF16=1 sends bf16 buckets, REHEARSE_RESERVE_CUS=k keeps k
compute units free of the persistent GEMM kernels; the watchdog prints where a stalled run stopped."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
faulthandler.dump_traceback_later(50, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])


def log(msg):
    print(f"[rank {rank} {time.strftime('%X')}] {msg}", flush=True)


torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(os.environ.get("VITMI_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
log("init ok")
on_device = "nccl" in str(dist.get_backend()).lower()
cdev = dev if on_device else torch.device("cpu")      # where this backend's collectives run
t = torch.full((1000,), float(rank + 1), device=cdev)
dist.all_reduce(t)
w = dist.all_reduce(t, async_op=True)
w.wait()
if on_device:
    torch.cuda.synchronize()
log(f"plain all_reduce ok: {t[0].item()}")

from visiontransformer_amd import ViT  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402
from visiontransformer_amd.parallel import GradReducer, broadcast_parameters  # noqa: E402

torch.manual_seed(1001 + rank)            # different initial weights: the broadcast must equalise them
if os.environ.get("REHEARSE_DEEP"):
    from visiontransformer_amd import ViT_mask
    m = ViT_mask.VIT(32, 4, 192, 10, 12, 12, 512).to(dev).train()
    log(f"deep model built: {len(list(m.parameters()))} parameter tensors")
else:
    m = ViT.VIT(32, 4, 64, 10, 2, 4, 128).to(dev).train()
broadcast_parameters(m, src=0)
log("broadcast ok")
red = GradReducer(m, bucket_mb=float(os.environ.get("REHEARSE_BUCKET_MB", "0.05")),
                  dtype=torch.bfloat16 if os.environ.get("REHEARSE_BF16") else torch.float32,
                  reserve_cus=int(os.environ.get("REHEARSE_RESERVE_CUS", "0")))
log(f"reducer: {red.info()}")
if os.environ.get("REHEARSE_TRACE"):
    import threading
    _orig = dist.all_reduce

    def traced(t, *a, **k):
        log(f"all_reduce numel {t.numel()} async {k.get('async_op', False)} thread {threading.current_thread().name}")
        return _orig(t, *a, **k)
    dist.all_reduce = traced
    import visiontransformer_amd.parallel as _par
    _par.dist.all_reduce = traced
opt = FusedAdam(m.parameters(), lr=1e-3)
torch.manual_seed(1234 + rank)
x = torch.randn(8, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (8,), device=dev)
for step in range(3):
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(m(x) if not os.environ.get("REHEARSE_DEEP") else m(x, unk_mask=None), y)
    log(f"step {step} forward ok")
    loss.backward()
    log(f"step {step} backward ok")
    red.finish()
    log(f"step {step} finish ok")
    opt.step()
    torch.cuda.synchronize()
    log(f"step {step} loss {loss.item():.5f}")
# all ranks must hold identical parameters after identical averaged updates
flat = torch.cat([p.detach().flatten() for p in m.parameters()]).to(cdev)
ref = flat.clone()
dist.broadcast(ref, src=0)
log(f"max parameter difference to rank 0: {(flat - ref).abs().max().item():.3e}")
dist.barrier()
log("done")
