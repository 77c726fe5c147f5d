#!/usr/bin/env python3
"""Which tensors sit above the 1e-4 / 2e-2 parity bars in the timm-style and DP-SGD tests, and by which measure:
max|a-b|/max|b| (the tests' definition) and the rms ratio."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err  # noqa: E402
from oracle import timm_vit_oracle, vit_oracle  # noqa: E402
import test_gpu_dpsgd as td  # noqa: E402
from test_gpu_timm_vit import make  # noqa: E402

DEV = "cuda:0"


def rms(a, b):
    a, b = a.detach().double().cpu(), b.double()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item()


def report(tag, named, ref, tol):
    rows = sorted(((rel_err(p.grad, ref[k]), rms(p.grad, ref[k]), k, float(ref[k].abs().max())) for k, p in named), reverse=True)
    over = [r for r in rows if r[0] >= tol]
    print(f"{tag}: worst {rows[0][0]:.3e} ({rows[0][2]}), {len(over)} tensors over {tol:g}")
    for r in rows[:4]:
        print(f"     max-rel {r[0]:.3e}  rms-rel {r[1]:.3e}  max|ref| {r[3]:.3e}  {r[2]}")


for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
    for masked in (False, True):
        for pool in ("token", "avg"):
            m, params, args = make(dtype, pool)
            g = torch.Generator().manual_seed(3)
            x = torch.randn(6, 3, 32, 32, generator=g)
            y = torch.randint(0, 10, (6,), generator=g)
            unk = None
            if masked:
                unk = torch.zeros(1, 65, dtype=torch.long)
                unk[0, torch.randperm(64, generator=g)[:32] + 1] = 1
            _, _, ref = timm_vit_oracle.loss_and_grads(params, x, y, args["num_heads"], unk_mask=unk, global_pool=pool)
            m.train()
            torch.nn.functional.cross_entropy(m(x.to(DEV), unk.to(DEV) if unk is not None else None), y.to(DEV)).backward()
            report(f"timm {dtype} masked={masked} pool={pool}", list(m.named_parameters()), ref, tol)
    from visiontransformer_amd import ViT, dpsgd
    for geom, kw in td.GEOMETRIES.items():
        okw = {k: v for k, v in kw.items() if k != "pool"}
        params = vit_oracle.init_params(masked=False, seed=3, **okw)
        gen = torch.Generator().manual_seed(5)
        x = torch.randn(5, 3, kw["image_size"], kw["image_size"], generator=gen)
        y = torch.randint(0, kw["num_classes"], (5,), generator=gen)
        ref_norms, ref = td.oracle_per_sample(vit_oracle.loss_and_grads, params, x, y, kw["heads"], pool=kw.get("pool", "cls"))
        m = ViT.VIT(**kw)
        m.load_state_dict(params)
        m = m.to(DEV).set_compute_dtype(dtype).train()
        _, norms = dpsgd.clipped_noised_gradients(m, x.to(DEV), y.to(DEV), l2_norm_clip=td.CLIP, noise_multiplier=0.0)
        print(f"dpsgd {geom} {dtype}: norms err {rel_err(norms, ref_norms):.3e}")
        report(f"dpsgd {geom} {dtype}", list(m.named_parameters()), ref, tol)
    m, params, args = make(dtype)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (4,), generator=gen)
    ref_norms, ref = td.oracle_per_sample(timm_vit_oracle.loss_and_grads, params, x, y, args["num_heads"])
    m.train()
    _, norms = dpsgd.clipped_noised_gradients(m, x.to(DEV), y.to(DEV), l2_norm_clip=td.CLIP, noise_multiplier=0.0, unk_mask=None)
    print(f"dpsgd timm {dtype}: norms err {rel_err(norms, ref_norms):.3e}")
    report(f"dpsgd timm {dtype}", list(m.named_parameters()), ref, tol)
