"""world_size-2 gloo test of the bucketed gradient exchange (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd.parallel import GradReducer, broadcast_parameters
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    for p in model[5].parameters():                     # a branch that never takes part in the graph
        pass
    broadcast_parameters(model, src=0)
    red = GradReducer(model, bucket_mb=0.0002)          # tiny buckets: several collectives per step
    assert len(red.buckets) > 2
    torch.manual_seed(7)
    x_all = torch.randn(6, 8)
    y_all = torch.randn(6, 4)
    xs, ys = x_all[rank * 3:(rank + 1) * 3], y_all[rank * 3:(rank + 1) * 3]
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        loss = ((model[:5](xs) - ys) ** 2).mean()        # model[5] unused -> zero gradient slots
        loss.backward()
        red.finish()
    grads = [p.grad.clone() for p in model.parameters()]
    if rank == 0:
        ref = torch.nn.Sequential(*[type(m)(*(() if isinstance(m, torch.nn.Tanh) else (m.in_features, m.out_features)))
                                    for m in model])
        ref.load_state_dict(model.state_dict())
        ((ref[:5](x_all) - y_all) ** 2).mean().backward()
        ok = True
        for g, p in zip(grads, ref.parameters()):
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            ok = ok and torch.allclose(g, want, atol=1e-6, rtol=1e-5)
        out.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_single_process_large_batch():
    """K ranks with per-rank batch b and mean loss == one process with batch K*b (the invariant
    that pins the data-parallel path)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    ok = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


class _NoWeightGrad(torch.autograd.Function):
    """y = x * w whose backward hands the weight nothing (what the per-sample-norm pass of DP-SGD does)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(w)
        return x * w

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return g * w, None


def _worker_undefined(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd.parallel import GradReducer
    torch.manual_seed(1)
    lin = torch.nn.Linear(4, 4)
    w = torch.nn.Parameter(torch.ones(4))
    model = torch.nn.ParameterList([w, *lin.parameters()])
    red = GradReducer(model, bucket_mb=0.00001, average=False)
    x = torch.full((2, 4), float(rank + 1))
    # a backward pass in which w's accumulation node runs with an undefined gradient: no arrival, no collective
    _NoWeightGrad.apply(x.clone().requires_grad_(True), w).sum().backward()
    assert w.grad is None and all(b.work is None for b in red.buckets)
    # the real pass: every parameter arrives, sums over the two ranks
    (lin(x) * w).sum().backward()
    red.finish()
    ok = torch.allclose(w.grad, sum(lin(torch.full((2, 4), float(r + 1))).sum(0) for r in range(world)).detach())
    if rank == 0:
        out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_ignores_undefined_gradient_arrivals():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_undefined, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def _worker_divergent(rank, world, port, out):
    """Each rank's backward pass skips a DIFFERENT branch of the model, with buckets so small that every parameter
    is its own collective: the ranks complete their buckets in different orders and leave different ones for
    finish(), yet must issue the same sequence of collectives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd.parallel import GradReducer, broadcast_parameters
    torch.manual_seed(3)
    trunk = torch.nn.Linear(6, 6)
    heads = torch.nn.ModuleList([torch.nn.Linear(6, 3), torch.nn.Linear(6, 3)])
    model = torch.nn.ModuleList([trunk, heads])
    broadcast_parameters(model, src=0)
    red = GradReducer(model, bucket_mb=0.00001)
    assert len(red.buckets) == 6
    torch.manual_seed(11)
    x_all = torch.randn(2, 4, 6)
    ok = True
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        heads[rank](torch.tanh(trunk(x_all[rank]))).pow(2).mean().backward()     # rank r uses head r only
        red.finish()
        # reference: the mean over ranks of each rank's own loss
        ref = [torch.nn.Linear(6, 6), torch.nn.ModuleList([torch.nn.Linear(6, 3), torch.nn.Linear(6, 3)])]
        ref = torch.nn.ModuleList(ref)
        ref.load_state_dict(model.state_dict())
        sum(ref[1][r](torch.tanh(ref[0](x_all[r]))).pow(2).mean() for r in range(world)).div(world).backward()
        for p, q in zip(model.parameters(), ref.parameters()):
            ok = ok and torch.allclose(p.grad, q.grad if q.grad is not None else torch.zeros_like(q), atol=1e-6)
    # gradient accumulation: two local backward passes, one exchange
    for p in model.parameters():
        p.grad = None
    with red.no_sync():
        heads[0](torch.tanh(trunk(x_all[rank]))).sum().backward()
    assert all(b.work is None for b in red.buckets)
    heads[0](torch.tanh(trunk(x_all[rank]))).sum().backward()
    red.finish()
    ref.zero_grad()
    sum(2 * ref[1][0](torch.tanh(ref[0](x_all[r]))).sum() for r in range(world)).div(world).backward()
    ok = ok and torch.allclose(trunk.weight.grad, ref[0].weight.grad, atol=1e-5)
    # bf16 buckets: the same exchange with half the bytes, within bf16 rounding
    red.remove()
    red16 = GradReducer(model, bucket_mb=0.00001, dtype=torch.bfloat16)
    for p in model.parameters():
        p.grad = None
    heads[rank](torch.tanh(trunk(x_all[rank]))).pow(2).mean().backward()
    red16.finish()
    ref.zero_grad()
    sum(ref[1][r](torch.tanh(ref[0](x_all[r]))).pow(2).mean() for r in range(world)).div(world).backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        ok = ok and p.grad.dtype == torch.float32 and torch.allclose(p.grad, q.grad, atol=2e-2 * float(q.grad.abs().max()))
    assert red16.info()["bucket_bytes"] == [2 * p.numel() for p in reversed(list(model.parameters()))]
    if rank == 0:
        out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_unused_parameters_issue_the_same_collectives():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_divergent, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def _worker_no_sync(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiontransformer_amd.parallel import GradReducer, broadcast_parameters
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(6, 12), torch.nn.Tanh(), torch.nn.Linear(12, 3))
    broadcast_parameters(model, src=0)
    red = GradReducer(model, bucket_mb=0.0001)
    torch.manual_seed(11)
    x_all, y_all = torch.randn(8, 6), torch.randn(8, 3)
    xs, ys = x_all[rank * 4:(rank + 1) * 4], y_all[rank * 4:(rank + 1) * 4]
    ok = True
    # (1) finish() inside no_sync() is an error, not a silent exchange of zeros
    with red.no_sync():
        ((model(xs[:2]) - ys[:2]) ** 2).sum().backward()
        try:
            red.finish()
            ok = False
        except RuntimeError as e:
            ok = ok and "no_sync" in str(e)
    # (2) a step made ONLY of no_sync backward passes: finish() afterwards sends what accumulated locally
    with red.no_sync():
        ((model(xs[2:]) - ys[2:]) ** 2).sum().backward()
    red.finish()
    got = [p.grad.clone() for p in model.parameters()]
    # (3) the next ordinary step starts from fresh gradients although p.grad are views of the buckets
    for p in model.parameters():
        p.grad = None
    ((model(xs) - ys) ** 2).sum().backward()
    red.finish()
    got2 = [p.grad.clone() for p in model.parameters()]
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(6, 12), torch.nn.Tanh(), torch.nn.Linear(12, 3))
        ref.load_state_dict(model.state_dict())
        (((ref(x_all) - y_all) ** 2).sum() / world).backward()
        for g, g2, p in zip(got, got2, ref.parameters()):
            ok = ok and torch.allclose(g, p.grad, atol=1e-6, rtol=1e-5) and torch.allclose(g2, p.grad, atol=1e-6, rtol=1e-5)
        out.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_no_sync_accumulation_is_sent_by_the_next_finish():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_no_sync, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True
