#!/usr/bin/env python3
"""A/B of process-wide kernel knobs on the whole ViT-B/16 training step, inside ONE process on ONE box (boxes differ
by several percent): the configurations take turns, `rounds` times `steps` steps each, no launch timers.
usage: ab_step.py "name:knob=value,knob=value" ...   knobs: group=<gm>, xspan=<ticks>, span=<ticks>, all=<0|1>,
classic_gelu=<0|1>"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from visiontransformer_amd import ViT  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

KW = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072, dim_head=64)


def apply(cfg):
    lib.vitmi_debug_set_nt_group(int(cfg.get("group", -1)))
    lib.vitmi_debug_set_nt_stagger(int(cfg.get("span", 0)), int(cfg.get("xspan", 0)), int(cfg.get("all", 0)))
    lib.vitmi_debug_set_gelu_pair(int(cfg.get("classic_gelu", 0)))
    for k, v in cfg.items():
        fn = getattr(lib, "vitmi_debug_set_" + k, None)
        if k not in ("group", "span", "xspan", "all", "classic_gelu"):
            if fn is None:
                raise SystemExit(f"unknown knob {k}")
            fn(int(v))


def main():
    cfgs = []
    for a in sys.argv[1:] or ["default:"]:
        name, _, rest = a.partition(":")
        cfgs.append((name, dict(kv.split("=") for kv in rest.split(",") if kv)))
    steps, rounds = int(os.environ.get("AB_STEPS", "10")), int(os.environ.get("AB_ROUNDS", "3"))
    dev = torch.device("cuda:0")
    torch.manual_seed(1001)
    model = ViT.VIT(**KW).to(dev).train()
    opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)
    x = torch.randn(256, 3, 224, 224, device=dev)
    y = torch.randint(0, 100, (256,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        opt.step()

    for _ in range(5):
        step()
    res = {n: [] for n, _ in cfgs}
    for r in range(rounds):
        for name, cfg in cfgs:
            apply(cfg)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
    if os.environ.get("AB_FAMILIES"):
        # per-family GPU time of each configuration (launch timers on every family over two steps)
        from visiontransformer_amd import _lib
        for name, cfg in cfgs:
            apply(cfg)
            step()
            torch.cuda.synchronize()
            lib.vitmi_prof_select(-1)
            lib.vitmi_prof_reset()
            lib.vitmi_prof_enable(1)
            step()
            step()
            torch.cuda.synchronize()
            lib.vitmi_prof_enable(0)
            fams = []
            for fam in range(_lib.FAM_COUNT):
                ms, work, n = _lib.prof_read(fam)
                if n:
                    fams.append((ms / 2, lib.vitmi_prof_name(fam).decode()))
            print(name, " ".join(f"{nm.split('|')[-1][:28]}={ms:.3f}" for ms, nm in sorted(fams, reverse=True)[:9]), flush=True)
    for name, _ in cfgs:
        v = res[name]
        print(f"{name:24s} ms/step: " + " ".join(f"{t:7.3f}" for t in v) + f"   min {min(v):7.3f}  -> {256 / min(v) * 1e3:7.0f} img/s",
              flush=True)
    apply({})


main()
