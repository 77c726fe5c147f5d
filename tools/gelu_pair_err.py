#!/usr/bin/env python3
"""Gradient error of the bf16 model against the reference fixtures with the two epilogue pairs of the MLP block
(vitmi_debug_set_gelu_pair): worst and rms-over-tensors relative error per fixture."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Golden, VIT_CASES, rel_err  # noqa: E402
import visiontransformer_amd  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402

for case in VIT_CASES:
    g = Golden(case)
    line = f"{case:14s}"
    for classic in (1, 0):
        lib.vitmi_debug_set_gelu_pair(classic)
        m = getattr(visiontransformer_amd, g.module).VIT(**g.ctor_kwargs())
        m.load_state_dict(g.params, strict=True)
        m = m.to("cuda:0").set_compute_dtype(torch.bfloat16).train()
        if g.masked:
            m.train_method = "mask"
        x = g.x.to("cuda:0")
        um = g.unk_mask.to("cuda:0") if g.unk_mask is not None else None
        logits = m(x, unk_mask=um) if g.module != "ViT" else m(x)
        torch.nn.CrossEntropyLoss()(logits, g.y.to("cuda:0")).backward()
        errs = sorted(((rel_err(p.grad, g.grads[k]), k) for k, p in m.named_parameters()
                       if g.grads[k].abs().max() > 0), reverse=True)
        line += f" | {'classic' if classic else 'deriv  '} worst {errs[0][0]:.4f} ({errs[0][1][-28:]}) 2nd {errs[1][0]:.4f} " \
                f"mean {sum(e for e, _ in errs) / len(errs):.4f}"
    print(line, flush=True)
lib.vitmi_debug_set_gelu_pair(0)
