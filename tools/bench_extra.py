#!/usr/bin/env python3
"""Secondary datapoints of the measurement plan (one GPU):
  export  : cifar100 geometry (ViT.VIT(32,4,192,100,12,12,512), batch 256, bf16): training step without
            export, and forward with the attention-probability export on (hooks collecting 12 x
            [256,12,65,65]).
  dpsgd   : DP-SGD step in its literal schedule -- 256 passes of batch 1, each gradient clipped to
            l2_norm_clip = 1.0, summed, Gaussian noise (noise_multiplier = 1.0) added, divided by 256
            (the intended algorithm of defence_method/DPSGD.py:97-146; that loop as written keeps only
            the last sample's gradient), then Adam.  Model: ViT-B/16 geometry unless --small.
Prints one JSON line per datapoint."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

DEV = torch.device("cuda", 0)


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def export_points():
    torch.manual_seed(1001)
    m = ViT.VIT(32, 4, 192, 100, 12, 12, 512).to(DEV)
    opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)
    x = torch.randn(256, 3, 32, 32, device=DEV)
    y = torch.randint(0, 100, (256,), device=DEV)

    def train():
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x), y).backward()
        opt.step()

    m.train()
    dt = timed(train, 5, 20)
    print(json.dumps({"point": "cifar100 bf16 train step, no export", "images_per_s": 256 / dt, "ms": dt * 1e3}))
    m.eval()
    seen = []
    for name, mod in m.named_modules():
        if name.endswith("attn_drop"):
            mod.register_forward_hook(lambda mm, i, o: seen.append(o))

    def fwd():
        seen.clear()
        with torch.no_grad():
            m(x)

    dt = timed(fwd, 5, 20)
    nbytes = sum(t.numel() * t.element_size() for t in seen)
    print(json.dumps({"point": "cifar100 bf16 forward with probability export (12 x [256,12,65,65])",
                      "images_per_s": 256 / dt, "ms": dt * 1e3, "export_bytes_per_image": nbytes / 256}))


def dpsgd_point(small, batched_only=False):
    torch.manual_seed(1001)
    kw = (dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=12, heads=12, mlp_dim=512) if small else
          dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072,
               dim_head=64))
    m = ViT.VIT(**kw).to(DEV).train()
    params = [p for p in m.parameters()]
    opt = FusedAdam(params, lr=1e-3, betas=(0.9, 0.99))
    B, clip, sigma = 256, 1.0, 1.0
    x = torch.randn(B, 3, kw["image_size"], kw["image_size"], device=DEV)
    y = torch.randint(0, kw["num_classes"], (B,), device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(1234)

    def step():
        acc = [torch.zeros_like(p) for p in params]
        for i in range(B):
            for p in params:
                p.grad = None
            torch.nn.functional.cross_entropy(m(x[i:i + 1]), y[i:i + 1]).backward()
            grads = [p.grad for p in params]
            norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
            scale = torch.clamp(clip / (norm + 1e-6), max=1.0)
            torch._foreach_add_(acc, torch._foreach_mul(grads, scale))
        for p, a in zip(params, acc):
            p.grad = (a + sigma * clip * torch.randn(a.shape, device=DEV, generator=gen)) / B
        opt.step()

    from visiontransformer_amd import dpsgd

    def batched():
        dpsgd.clipped_noised_gradients(m, x, y, clip, sigma, generator=gen)
        opt.step()

    dt = timed(batched, 2, 5)
    print(json.dumps({"point": f"DP-SGD batched (one forward, norm backward, clipped backward), "
                               f"{'cifar' if small else 'ViT-B/16'} geometry", "images_per_s": B / dt, "ms": dt * 1e3}),
          flush=True)
    if batched_only:
        return
    dt = timed(step, 1, 2)
    print(json.dumps({"point": f"DP-SGD literal schedule, 256 x (B=1), {'cifar' if small else 'ViT-B/16'} geometry",
                      "images_per_s": B / dt, "ms": dt * 1e3}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["export", "dpsgd", "all"])
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--batched-only", action="store_true")
    a = ap.parse_args()
    if a.what in ("export", "all"):
        export_points()
    if a.what in ("dpsgd", "all"):
        dpsgd_point(a.small, a.batched_only)
