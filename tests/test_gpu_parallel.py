"""The data-parallel step through RCCL on the GPU: one rank (a one-GPU box cannot hold more), launched the way
the benchmark is launched for N > 1 (torch.distributed.run, backend "nccl").  It drives broadcast_parameters,
the gradient hooks, the packed bucket copy, the asynchronous all_reduce on RCCL's stream and finish() on real
device streams; the K-rank arithmetic is covered by the gloo test on the CPU."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_step_runs_and_learns():
    env = dict(os.environ, VITMI_DIST_BACKEND="nccl", REHEARSE_DEEP="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tools", "dist_rehearsal.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    losses = [float(x) for x in re.findall(r"step \d+ loss ([0-9.]+)", out)]
    assert len(losses) == 3 and losses[-1] < losses[0], out[-2000:]
    assert "max parameter difference to rank 0: 0.000e+00" in out


def test_torch_ddp_wraps_the_module_as_the_trainer_does():
    """trainer.py:184: DistributedDataParallel(model, find_unused_parameters=True) around ViT_mask.VIT, RCCL, one rank."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "tools", "ddp_wrap_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "ddp wrap ok" in out, out[-3000:]


def test_data_parallel_wrapper_on_one_device():
    """nn.DataParallel around the drop-in module (one visible device: the wrapper scatters, calls the module and
    gathers on cuda:0): same logits and gradients as the bare module."""
    import torch
    from visiontransformer_amd import ViT
    kw = dict(image_size=32, patch_size=4, embed_dim=64, num_classes=10, depth=2, heads=4, mlp_dim=128)
    torch.manual_seed(0)
    m = ViT.VIT(**kw).to("cuda:0").train()
    x = torch.randn(8, 3, 32, 32, device="cuda:0")
    y = torch.randint(0, 10, (8,), device="cuda:0")
    torch.nn.functional.cross_entropy(m(x), y).backward()
    want = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    dp = torch.nn.DataParallel(m, device_ids=[0])
    out = dp(x)
    assert torch.equal(out, m(x))
    torch.nn.functional.cross_entropy(out, y).backward()
    assert all(torch.equal(p.grad, want[k]) for k, p in m.named_parameters())


def test_dpsgd_through_the_reducer_on_rccl():
    """Batched DP-SGD with the gradient reducer (sum over ranks, one noise draw, division by the global batch): one
    rank on RCCL gives bit for bit the gradient of the step without a reducer."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tools", "dpsgd_dist_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "dpsgd dist ok" in out, out[-3000:]
    assert "with / without the reducer on one rank: 0.000e+00" in out
