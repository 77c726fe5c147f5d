#!/usr/bin/env python3
"""Both routes of dpsgd.clipped_noised_gradients (clipped replay of the norm pass / ordinary second backward) against
the one-image-at-a-time oracle: worst relative error per route, and the LayerNorm weights of the timm-style model."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import rel_err  # noqa: E402
from test_gpu_dpsgd import CLIP, GEOMETRIES, oracle_per_sample  # noqa: E402
from oracle import timm_vit_oracle, vit_oracle  # noqa: E402
from visiontransformer_amd import ViT, dpsgd  # noqa: E402

DEV = "cuda:0"
for geom, kw in GEOMETRIES.items():
    okw = {k: v for k, v in kw.items() if k != "pool"}
    params = vit_oracle.init_params(masked=False, seed=3, **okw)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(5, 3, kw["image_size"], kw["image_size"], generator=gen)
    y = torch.randint(0, kw["num_classes"], (5,), generator=gen)
    ref_norms, ref = oracle_per_sample(vit_oracle.loss_and_grads, params, x, y, kw["heads"], pool=kw.get("pool", "cls"))
    for dtype in (torch.float32, torch.bfloat16):
        res = {}
        for reuse in (True, False):
            m = ViT.VIT(**kw)
            m.load_state_dict(params)
            m = m.to(DEV).set_compute_dtype(dtype).train()
            dpsgd.clipped_noised_gradients(m, x.to(DEV), y.to(DEV), CLIP, 0.0, reuse_norm_pass=reuse)
            res[reuse] = {k: p.grad.clone() for k, p in m.named_parameters()}
            worst = max((rel_err(g, ref[k]), k) for k, g in res[reuse].items())
            print(f"{geom} {dtype} reuse={reuse}: worst {worst[0]:.3e} {worst[1]}")
        print(f"   routes differ by {max(rel_err(res[True][k], res[False][k]) for k in ref):.3e}")

from test_gpu_timm_vit import make  # noqa: E402
for dtype in (torch.float32, torch.bfloat16):
    for reuse in (True, False):
        m, params, args = make(dtype)
        gen = torch.Generator().manual_seed(9)
        x = torch.randn(4, 3, 32, 32, generator=gen)
        y = torch.randint(0, 10, (4,), generator=gen)
        ref_norms, ref = oracle_per_sample(timm_vit_oracle.loss_and_grads, params, x, y, args["num_heads"])
        m.train()
        dpsgd.clipped_noised_gradients(m, x.to(DEV), y.to(DEV), CLIP, 0.0, unk_mask=None, reuse_norm_pass=reuse)
        errs = sorted(((rel_err(p.grad, ref[k]), k) for k, p in m.named_parameters()), reverse=True)
        print(f"timm {dtype} reuse={reuse}: worst three {[(round(e, 5), k) for e, k in errs[:3]]}")
