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


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def launch(script, nproc=1, **env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env)
    cmd = [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", free_port(), os.path.join(ROOT, "tools", script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    return r.returncode, r.stdout + r.stderr


def test_single_rank_rccl_step_runs_and_learns():
    rc, out = launch("dist_rehearsal.py", VITMI_DIST_BACKEND="nccl", REHEARSE_DEEP="1")
    assert rc == 0, out[-3000:]
    losses = [float(x) for x in re.findall(r"step \d+ loss ([0-9.]+)", out)]
    assert len(losses) == 3 and losses[-1] < losses[0], out[-2000:]
    assert "max parameter difference to rank 0: 0.000e+00" in out


def test_single_rank_rccl_step_with_bf16_buckets_and_reserved_compute_units():
    """The two switches for the multi-GPU run: gradients travel as bf16 (half the bytes on the links) and the
    persistent GEMM kernels leave 16 compute units to the collective; one rank on RCCL, the step still learns."""
    rc, out = launch("dist_rehearsal.py", VITMI_DIST_BACKEND="nccl", REHEARSE_DEEP="1", REHEARSE_BF16="1",
                     REHEARSE_RESERVE_CUS="16")
    assert rc == 0, out[-3000:]
    assert "'dtype': 'bfloat16'" in out and "'reserved_cus': 16" in out
    losses = [float(x) for x in re.findall(r"step \d+ loss ([0-9.]+)", out)]
    assert len(losses) == 3 and losses[-1] < losses[0], out[-2000:]


def test_two_ranks_on_one_gpu_through_host_staged_gloo():
    """Two processes drive the HIP modules on the one GPU and exchange gradients through gloo on host copies of the
    buckets (no device tensor reaches gloo): the broadcast equalises the differently initialised replicas and three
    averaged steps leave bit-identical parameters on both ranks."""
    rc, out = launch("dist_rehearsal.py", nproc=2, VITMI_DIST_BACKEND="gloo")
    assert rc == 0, out[-3000:]
    assert out.count("max parameter difference to rank 0: 0.000e+00") == 2, out[-2000:]
    l0 = re.findall(r"\[rank 0 [^\]]*\] step \d+ loss ([0-9.]+)", out)
    l1 = re.findall(r"\[rank 1 [^\]]*\] step \d+ loss ([0-9.]+)", out)
    assert len(l0) == 3 and len(l1) == 3 and l0 != l1          # different data per rank


def test_torch_ddp_wraps_the_module_as_the_trainer_does():
    """trainer.py:184: DistributedDataParallel(model, find_unused_parameters=True) around ViT_mask.VIT, RCCL, one rank."""
    rc, out = launch("ddp_wrap_check.py")
    assert rc == 0 and "ddp wrap ok" in out, out[-3000:]


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
    rc, out = launch("dpsgd_dist_check.py")
    assert rc == 0 and "dpsgd dist ok" in out, out[-3000:]
    assert "with / without the reducer on one rank: 0.000e+00" in out
