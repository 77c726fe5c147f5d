import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

VIT_CASES = ["cifar_dh8", "dh64_mean", "n197_dh64", "identity_out", "mask_dh8", "mask_none", "mask_dh64",
             "avg_mask", "avg_none", "ape_none"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """One vit_<case>.npz fixture split into params / grads / cfg / tensors."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, f"vit_{name}.npz"), allow_pickle=False)
        self.name = name
        self.params, self.grads, self.cfg, self.probs = {}, {}, {}, {}
        for k in z.files:
            if k.startswith("p/"):
                self.params[k[2:]] = torch.from_numpy(z[k])
            elif k.startswith("g/"):
                self.grads[k[2:]] = torch.from_numpy(z[k])
            elif k.startswith("cfg/"):
                v = z[k]
                self.cfg[k[4:]] = str(v) if v.dtype.kind in "US" else int(v)
            elif k.startswith("probs/"):
                self.probs[int(k[6:])] = torch.from_numpy(z[k])
        self.x = torch.from_numpy(z["x"])
        self.y = torch.from_numpy(z["y"])
        self.logits = torch.from_numpy(z["logits"])
        self.loss = torch.from_numpy(z["loss"])
        self.feat = torch.from_numpy(z["feat"])
        self.unk_mask = torch.from_numpy(z["unk_mask"]) if "unk_mask" in z.files else None
        self.masked = bool(self.cfg.pop("masked"))
        # reference module the fixture was captured from (older fixtures: ViT or ViT_mask)
        self.module = self.cfg.pop("module", "ViT_mask" if self.masked else "ViT")

    def ctor_kwargs(self):
        return dict(self.cfg)


@pytest.fixture(params=VIT_CASES)
def golden(request):
    return Golden(request.param)


def rel_err(a, b):
    """max|a-b| / max|b| -- the 'relative' of the fp32 1e-4 / bf16 2e-2 parity bars."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.abs().max().item()
    if denom == 0.0:
        return (a - b).abs().max().item()
    return (a - b).abs().max().item() / denom
