"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce overlapped with
backward (the exchange torch DDP performs for trainer.py:184 in the reference).

Gradients are packed into a few large flat buckets in reverse parameter order (the last
layers' gradients are final first).  A bucket's all-reduce is issued, asynchronously, from the
hook of the last parameter that lands in it, so the collective (RCCL over xGMI for the "nccl"
backend) runs under the rest of backward.  Collectives are issued STRICTLY in bucket order on
every rank -- a bucket that completes early waits for the ones before it, and finish() issues
whatever is left in the same order -- so ranks whose backward passes touch different parameter
sets (find_unused_parameters behaviour, a data-dependent branch) still enqueue identical
sequences of collectives.  finish() waits for them and hands every parameter its averaged
gradient.  Works on any backend torch.distributed offers; a backend without device collectives
(gloo with ROCm tensors) is fed through host copies of the buckets.
"""
import contextlib

import torch
import torch.distributed as dist


def _device_collectives(group=None):
    """True when the group's backend runs collectives on device tensors itself (nccl = RCCL on ROCm)."""
    return "nccl" in str(dist.get_backend(group)).lower()


def broadcast_parameters(module, src=0, group=None):
    """Rank `src`'s parameters and buffers everywhere (what DDP's constructor does).  The copy goes through
    Tensor.copy_, so every parameter's version counter moves and compute copies cached from the old values
    (visiontransformer_amd.ops) are rebuilt; the cache is dropped as well for tensors whose storage was swapped."""
    with torch.no_grad():
        by_kind = {}
        for t in list(module.parameters()) + list(module.buffers()):
            by_kind.setdefault((t.dtype, t.device), []).append(t)
        for (_, device), tensors in by_kind.items():     # one collective per dtype/device, not one per tensor
            flat = torch.cat([t.detach().reshape(-1) for t in tensors])
            if flat.is_cuda and not _device_collectives(group):
                host = flat.cpu()
                dist.broadcast(host, src=src, group=group)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    try:
        from . import ops
        ops.clear_cast_cache()
    except ImportError:          # the CPU-only tests import this module without the HIP library
        pass


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
        self.ready = False
        self.work = None
        self.host = None

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)


class GradReducer:
    """Attach to a module's parameters; call finish() after backward, before the optimizer step.

    bucket_mb: bucket size in MiB of fp32 gradients.  xGMI links are point to point, so a ring
    all-reduce is bound by one link: few large buckets amortise the per-collective latency.
    dtype: what travels.  torch.float32 (default) sends the gradients as they are (343 MB per step for
    ViT-B/16); torch.bfloat16 halves the bytes on the links: gradients are rounded to bf16 when packed, summed by
    the collective in bf16, and handed back as fp32.
    reserve_cus: compute units the persistent GEMM kernels leave free while this reducer is attached (a multiple
    of 8; 0 = none).  Those kernels otherwise hold every CU with one 512-thread workgroup, so a collective's
    workgroups could only start when a GEMM launch drains.
    """

    def __init__(self, module, group=None, bucket_mb=64, average=True, dtype=torch.float32, reserve_cus=0):
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs torch.distributed to be initialised")
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("GradReducer sends float32 or bfloat16")
        self.group = group
        self.world = dist.get_world_size(group)
        self.average = average
        self.dtype = dtype
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
        self._stage = bool(self.buckets) and self.buckets[0].flat.is_cuda and not _device_collectives(group)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self._seen = set()
        self._dirty = set()         # parameters that accumulated a gradient under no_sync() since the last exchange
        self._next = 0              # first bucket whose collective has not been issued this step
        self._sync = True
        self.reserve_cus = int(reserve_cus)
        if self.reserve_cus:
            self._reserve(self.reserve_cus)
        # timeline of the exchange (device tensors only): events of the last finished step, see timeline()
        self.record_timeline = False
        self._ev_bwd_start = None
        self._ev_issue = {}
        self._last = None

    @staticmethod
    def _reserve(k):
        from ._lib import check, lib
        check(lib.vitmi_set_reserved_cus(int(k)), "vitmi_set_reserved_cus")

    def info(self):
        """What one step puts on the links: for the benchmark's JSON line."""
        esz = 2 if self.dtype == torch.bfloat16 else 4
        return {"buckets": len(self.buckets), "bucket_bytes": [b.flat.numel() * esz for b in self.buckets],
                "dtype": str(self.dtype).replace("torch.", ""), "world": self.world, "reserved_cus": self.reserve_cus,
                "backend": str(dist.get_backend(self.group))}

    def _close(self, params):
        b = _Bucket(list(params), params[0].device, self.dtype)
        for i, p in enumerate(params):
            self._slot[p] = (b, i)
        self.buckets.append(b)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate into p.grad locally and exchange nothing (gradient accumulation, like
        DistributedDataParallel.no_sync); the first backward outside sends the accumulated gradients."""
        self._sync = False
        try:
            yield self
        finally:
            self._sync = True

    def _on_grad(self, p):
        if not self._sync:
            if p.grad is not None:
                self._dirty.add(p)  # its local sum lives in p.grad and is sent by the next exchange
            return
        if p.grad is None:      # the accumulation node ran with an undefined gradient (a backward pass that hands
            return              # this parameter nothing, e.g. the per-sample-norm pass of DP-SGD): not an arrival
        b, i = self._slot[p]
        if p in self._seen:  # second accumulation into the same gradient in one backward
            raise RuntimeError("GradReducer: a parameter received two gradients before finish(); accumulate "
                               "under no_sync() instead")
        self._seen.add(p)
        b.pending -= 1
        if b.pending == 0:
            b.ready = True
            self._launch_ready()

    def _launch_ready(self):
        """Issue the collectives of complete buckets, in bucket order, up to the first incomplete one."""
        while self._next < len(self.buckets) and self.buckets[self._next].ready:
            self._launch(self.buckets[self._next])
            self._next += 1

    def mark_backward_start(self):
        """Call right before loss.backward() when record_timeline is on: the origin of the step's timeline."""
        if self.record_timeline and self.buckets and self.buckets[0].flat.is_cuda:
            self._ev_bwd_start = torch.cuda.Event(enable_timing=True)
            self._ev_bwd_start.record()

    def timeline(self):
        """Of the last step finished with record_timeline on (synchronises the device): milliseconds from
        mark_backward_start() to the end of backward (`backward_ms`), from the end of backward until every bucket has
        been reduced and unpacked on the compute stream (`exposed_comm_ms`: what the step pays for the exchange beyond
        the backward pass it hides under), and per bucket when its collective was issued and when the compute stream
        was past its wait (`buckets`: [issue_ms, done_ms] from the same origin, in bucket order)."""
        if self._last is None:
            return None
        start, bwd_end, comm_done, issue, done = self._last
        torch.cuda.synchronize()
        t0 = start if start is not None else bwd_end
        return {"backward_ms": t0.elapsed_time(bwd_end) if start is not None else None,
                "exposed_comm_ms": bwd_end.elapsed_time(comm_done),
                "buckets": [[round(t0.elapsed_time(i), 3), round(t0.elapsed_time(d), 3)] for i, d in zip(issue, done)]}

    def _launch(self, b):
        if self.record_timeline and b.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._ev_issue[id(b)] = ev
        self._pack(b)
        if self._stage:             # no device collectives on this backend: reduce a host copy
            b.host = b.flat.float().cpu() if b.flat.dtype != torch.float32 else b.flat.cpu()
            b.work = dist.all_reduce(b.host, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _pack(self, b):
        """Gradients of a bucket into its flat buffer (rounded to the bucket's dtype): one multi-tensor copy, not
        one launch per tensor; slots of parameters that took no gradient are zeroed."""
        dst, src = [], []
        for i, p in enumerate(b.params):
            v = b.view(i)
            if (p in self._seen or p in self._dirty) and p.grad is not None:
                if p.grad.data_ptr() != v.data_ptr():
                    dst.append(v)
                    src.append(p.grad)
            else:
                v.zero_()
        if dst:
            torch._foreach_copy_(dst, src)

    def finish(self):
        """Wait for the exchange; leave averaged gradients in p.grad.  Parameters that took no
        gradient this step (unused in the graph) contribute zeros, like DDP with
        find_unused_parameters=True."""
        if not self._sync:
            raise RuntimeError("GradReducer.finish() inside no_sync(): leave the context first (the backward passes "
                               "inside it only accumulate locally)")
        scale = 1.0 / self.world if self.average else 1.0
        rec = self.record_timeline and bool(self.buckets) and self.buckets[0].flat.is_cuda
        if rec:
            bwd_end = torch.cuda.Event(enable_timing=True)
            bwd_end.record()
            done = []
        for b in self.buckets[self._next:]:     # incomplete buckets, in the order every rank uses
            self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            b.work.wait()
            if rec:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                done.append(ev)
            if b.host is not None:
                b.flat.copy_(b.host)
                b.host = None
            if scale != 1.0:
                b.flat.mul_(scale)
            if b.flat.dtype == torch.float32:
                # hand out views of the bucket instead of copies: the optimizer reads them before the next
                # backward refills the bucket (a parameter whose gradient is kept across steps then
                # accumulates in place inside the bucket, which the refill copies onto itself)
                for i, p in enumerate(b.params):
                    p.grad = b.view(i)
            else:
                dst, src = [], []
                for i, p in enumerate(b.params):
                    if p.grad is None or p.grad.dtype != p.dtype:
                        p.grad = torch.empty_like(p)
                    dst.append(p.grad)
                    src.append(b.view(i))
                torch._foreach_copy_(dst, src)
            b.pending = len(b.params)
            b.ready = False
            b.work = None
        if rec:
            comm_done = torch.cuda.Event(enable_timing=True)
            comm_done.record()
            self._last = (self._ev_bwd_start, bwd_end, comm_done, [self._ev_issue[id(b)] for b in self.buckets], done)
            self._ev_issue = {}
        self._next = 0
        self._seen.clear()
        self._dirty.clear()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        if self.reserve_cus:
            self._reserve(0)
            self.reserve_cus = 0
