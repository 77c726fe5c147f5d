#!/usr/bin/env python3
"""Time of one attention rollout (vitmi_rollout) against the same computation issued as the reference's torch ops
on the same device (yardstick only).  Algorithmic bytes = the exported probabilities, read once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import vit_rollout  # noqa: E402

dev = "cuda:0"


def torch_ops_rollout(attentions, discard_ratio):
    result = torch.eye(attentions[0].size(-1), device=dev)
    for attention in attentions:
        fused = attention.float().mean(axis=1)
        flat = fused.view(fused.size(0), -1)
        _, idx = flat.topk(int(flat.size(-1) * discard_ratio), -1, False)
        flat.scatter_(1, idx, 0)
        a = (fused + torch.eye(fused.size(-1), device=dev)) / 2
        a = a / a.sum(dim=-1)[..., None]
        result = torch.matmul(a, result)
    mask = result[:, 0, 1:]
    mask = mask / mask.max(-1)[0][..., None]
    return torch.softmax(mask, dim=-1)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (B, H, N, L, dt) in [(256, 12, 197, 12, torch.bfloat16), (256, 12, 65, 12, torch.bfloat16),
                         (128, 12, 197, 12, torch.float32)]:
    atts = [torch.softmax(torch.randn(B, H, N, N, device=dev), -1).to(dt) for _ in range(L)]
    nbytes = L * B * H * N * N * atts[0].element_size()
    ms = timed(lambda: vit_rollout.rollout(atts, 0.9, "mean"))
    ref = timed(lambda: torch_ops_rollout(atts, 0.9), n=2)
    print(f"B={B} H={H} N={N} L={L} {dt}: HIP {ms:.3f} ms ({nbytes / ms / 1e6:.0f} GB/s of probabilities), "
          f"torch ops {ref:.3f} ms, x{ref / ms:.1f}", flush=True)
