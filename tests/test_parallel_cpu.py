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
