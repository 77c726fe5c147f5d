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
