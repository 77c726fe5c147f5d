"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce overlapped with
backward (the exchange torch DDP performs for trainer.py:184 in the reference).

Gradients are packed into a few large flat buckets in reverse parameter order (the last
layers' gradients are final first).  A bucket's all-reduce is issued, asynchronously, from the
hook of the last parameter that lands in it, so the collective (RCCL over xGMI for the "nccl"
backend) runs under the rest of backward.  finish() waits for the collectives and hands every
parameter its averaged gradient.  Works on any backend torch.distributed offers (gloo on CPU
tensors is what the CPU tests use).
"""
import torch
import torch.distributed as dist


def broadcast_parameters(module, src=0, group=None):
    """Rank `src`'s parameters and buffers everywhere (what DDP's constructor does)."""
    with torch.no_grad():
        by_kind = {}
        for t in list(module.parameters()) + list(module.buffers()):
            by_kind.setdefault((t.dtype, t.device), []).append(t)
        for tensors in by_kind.values():           # one collective per dtype/device, not one per tensor
            flat = torch.cat([t.data.reshape(-1) for t in tensors])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in tensors:
                t.data.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


class _Bucket:
    def __init__(self, params, device, dtype):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=dtype, device=device)
        self.pending = len(params)
        self.work = None

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)


class GradReducer:
    """Attach to a module's parameters; call finish() after backward, before the optimizer step.

    bucket_mb: bucket size in MiB of fp32 gradients.  xGMI links are point to point, so a ring
    all-reduce is bound by one link: few large buckets amortise the per-collective latency.
    """

    def __init__(self, module, group=None, bucket_mb=64, average=True):
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs torch.distributed to be initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.average = average
        params = [p for p in module.parameters() if p.requires_grad]
        self.buckets = []
        self._slot = {}
        cur, cur_bytes = [], 0
        limit = bucket_mb * (1 << 20)
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= limit:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self._seen = set()

    def _close(self, params):
        b = _Bucket(list(params), params[0].device, torch.float32)
        for i, p in enumerate(params):
            self._slot[p] = (b, i)
        self.buckets.append(b)

    def _on_grad(self, p):
        if p.grad is None:      # the accumulation node ran with an undefined gradient (a backward pass that hands
            return              # this parameter nothing, e.g. the per-sample-norm pass of DP-SGD): not an arrival
        b, i = self._slot[p]
        if p in self._seen:  # second accumulation into the same gradient in one backward
            raise RuntimeError("GradReducer: a parameter received two gradients before finish()")
        self._seen.add(p)
        b.pending -= 1
        if b.pending == 0:
            self._pack(b)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _pack(self, b):
        """Gradients of a complete bucket into its flat buffer: one multi-tensor copy, not one launch per tensor."""
        dst, src = [], []
        for i, p in enumerate(b.params):
            if p in self._seen:
                v = b.view(i)
                if p.grad.data_ptr() != v.data_ptr():
                    dst.append(v)
                    src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)

    def finish(self):
        """Wait for the exchange; leave averaged gradients in p.grad.  Parameters that took no
        gradient this step (unused in the graph) contribute zeros, like DDP with
        find_unused_parameters=True."""
        scale = 1.0 / self.world if self.average else 1.0
        for b in self.buckets:
            if b.pending != 0:  # some parameters unused this step: zero their slots, then reduce
                for i, p in enumerate(b.params):
                    if p not in self._seen:
                        b.view(i).zero_()
                self._pack(b)
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for b in self.buckets:
            b.work.wait()
            if scale != 1.0:
                b.flat.mul_(scale)
            # hand out views of the bucket instead of copies: the optimizer reads them before the next
            # backward refills the bucket (a parameter whose gradient is kept across steps then
            # accumulates in place inside the bucket, which the refill copies onto itself)
            for i, p in enumerate(b.params):
                p.grad = b.view(i)
            b.pending = len(b.params)
            b.work = None
        self._seen.clear()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
