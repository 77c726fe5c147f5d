"""Fused Adam over fp32 master parameters (torch.optim.Adam semantics, the optimizer of
trainer.py:204-205) that refreshes the bf16 compute copies of the weights in the same pass."""
import torch

from . import ops
from ._lib import check, lib


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self.step_count += 1
        stream = torch.cuda.current_stream().cuda_stream
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            if not p.is_cuda:
                raise RuntimeError("FusedAdam runs on ROCm device tensors only")
            st = self.state.get(p)
            if st is None:
                st = (torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
                self.state[p] = st
            g = g.contiguous()
            hit = ops._cast_cache.get((id(p), torch.bfloat16))
            p16 = None
            if hit is not None and hit[0] == p._version and hit[1] == p.data_ptr():
                p16 = hit[2]
            # the transposed copy cannot be refreshed in the same pass: drop it, it is rebuilt on use
            ops._cast_cache.pop((id(p), torch.bfloat16, "T"), None)
            check(lib.vitmi_adam_step(p.data_ptr(), g.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                      p16.data_ptr() if p16 is not None else None, p.numel(), self.lr, self.betas[0],
                                      self.betas[1], self.eps, self.weight_decay, self.step_count, grad_scale, stream),
                  "vitmi_adam_step")
