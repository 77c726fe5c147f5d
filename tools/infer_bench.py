#!/usr/bin/env python3
"""ViT-B/16 224, batch 256, bf16, model.eval() under torch.no_grad(): the inference entry points (LayerNorm folded into
the qkv / fc1 products, no saved tensors) against the training forward, alternating in one process on one box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ViT, ops  # noqa: E402

KW = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072, dim_head=64)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(7)
model = ViT.VIT(**KW).to("cuda:0").eval()
x = torch.randn(B, 3, 224, 224, device="cuda:0")
res = {True: [], False: []}
outs = {}
with torch.no_grad():
    for rep in range(3):
        for on in (True, False):
            ops.set_inference_fusion(on)
            for _ in range(3):
                outs[on] = model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                model(x)
            torch.cuda.synchronize()
            res[on].append((time.perf_counter() - t0) / 10 * 1e3)
ops.set_inference_fusion(True)
err = ((outs[True] - outs[False]).abs().max() / outs[False].abs().max()).item()
for on in (False, True):
    t = min(res[on])
    print(f"inference entry points {'on ' if on else 'off'}: {t:.3f} ms per forward, {B / t * 1e3:.0f} images/s", flush=True)
print(f"logits on vs off: {err:.2e} relative")
import json  # noqa: E402
t = min(res[True])
gflop = 2 * 17.563137024                      # forward GFLOP per image (2 x MACs)
print(json.dumps({"metric": "inference images/sec ViT-B/16 224px (forward only, torch.no_grad)", "value": B / t * 1e3,
                  "unit": "images/s", "n_gpus": 1, "ms_per_step": t, "higher_is_better": True, "dtype": "bf16",
                  "data": "synthetic", "config": {"workload": f"vit_b16_224 forward, batch {B}", "global_batch": B,
                                                  "seq_len": 197},
                  "fwd_mfma_frac": B / t * 1e3 * gflop / 1e3 / 2500.0,
                  "training_forward_ms": min(res[False])}))
