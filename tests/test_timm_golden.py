"""The timm-style VisionTransformer with unk_mask against vectors of the reference's own class.

tests/golden/timm_*.npz were captured (tools/gen_golden.py, gen_timm_vit) by running the class definitions of the
reference's tim/models/vision_transformer.py, compiled out of the file together with the few vendored tim/layers
definitions their constructors call (the packages themselves cannot be imported here: torchvision and the pip
package timm are absent).  The oracle restatement is held to 2e-5 (forward) / 5e-5 (gradients) against them on the
CPU; the HIP model to fp32 1e-4 / bf16 2e-2 (max|a-b| / max|b|)."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import timm_vit_oracle

CASES = ["timm_token", "timm_token_mask", "timm_avg_mask", "timm_dh64_n197"]
DEV = "cuda:0"


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
        self.grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
        self.probs = [torch.from_numpy(z[f"probs/{i}"]) for i in range(sum(k.startswith("probs/") for k in z.files))]
        self.x, self.y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
        self.logits, self.loss, self.features = (torch.from_numpy(z[k]) for k in ("logits", "loss", "features"))
        self.unk = torch.from_numpy(z["unk_mask"]) if "unk_mask" in z.files else None
        self.heads, self.pool = int(z["cfg/heads"]), str(z["cfg/pool"])
        self.kw = ast.literal_eval(str(z["cfg/kw"]))


@pytest.mark.parametrize("name", CASES)
def test_oracle_against_reference_vectors(name):
    c = Case(name)
    probs = []
    feat = timm_vit_oracle.forward_features(c.params, c.x, c.heads, unk_mask=c.unk, collect_probs=probs)
    assert rel_err(feat, c.features) < 2e-5
    assert len(probs) == len(c.probs) and all(rel_err(a, b) < 2e-5 for a, b in zip(probs, c.probs))
    logits, loss, grads = timm_vit_oracle.loss_and_grads(c.params, c.x, c.y, c.heads, unk_mask=c.unk, global_pool=c.pool)
    assert rel_err(logits, c.logits) < 2e-5 and abs(loss.item() - c.loss.item()) < 2e-5
    assert set(grads) == set(c.grads)
    for k, g in c.grads.items():
        assert rel_err(grads[k], g) < 5e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("name", CASES)
def test_hip_model_against_reference_vectors(name, dtype, tol):
    from visiontransformer_amd.vision_transformer import VisionTransformer
    c = Case(name)
    m = VisionTransformer(**c.kw)
    m.load_state_dict(c.params, strict=True)                 # the reference class's own state_dict keys
    m = m.to(DEV).set_compute_dtype(dtype).train()
    seen = []
    hooks = [mod.register_forward_hook(lambda mm, i, o: seen.append(o.detach().float().cpu()))
             for n, mod in m.named_modules() if "attn_drop" in n]
    unk = c.unk.to(DEV) if c.unk is not None else None
    feat = m.forward_features(c.x.to(DEV), unk)
    logits = m.forward_head(feat)
    for h in hooks:
        h.remove()
    assert rel_err(feat, c.features) < tol and rel_err(logits, c.logits) < tol
    assert len(seen) == len(c.probs) and all(rel_err(a, b) < tol for a, b in zip(seen, c.probs))
    loss = torch.nn.functional.cross_entropy(logits, c.y.to(DEV))
    assert abs(loss.item() - c.loss.item()) < tol * max(1.0, abs(c.loss.item()))
    loss.backward()
    worst = max((rel_err(p.grad, c.grads[k]), k) for k, p in m.named_parameters())
    assert worst[0] < tol, worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_model_without_autograd_graph_against_reference_vectors(name):
    """model.eval() under torch.no_grad() (the evaluation and attack scripts' mode): qkv WITH bias and LayerNorm eps
    1e-6 through the folded products of vitmi_attn_block_infer / vitmi_ff_block_infer; logits, features and the
    probabilities a hook on attn_drop sees against the reference class's, bf16 bar."""
    from visiontransformer_amd.vision_transformer import VisionTransformer
    c = Case(name)
    m = VisionTransformer(**c.kw)
    m.load_state_dict(c.params, strict=True)
    m = m.to(DEV).set_compute_dtype(torch.bfloat16).eval()
    seen = []
    hooks = [mod.register_forward_hook(lambda mm, i, o: seen.append(o.detach().float().cpu()))
             for n, mod in m.named_modules() if "attn_drop" in n]
    unk = c.unk.to(DEV) if c.unk is not None else None
    with torch.no_grad():
        feat = m.forward_features(c.x.to(DEV), unk)
        n_probs = len(seen)
        logits = m(c.x.to(DEV), unk_mask=unk) if unk is not None else m(c.x.to(DEV))
    for h in hooks:
        h.remove()
    assert rel_err(feat, c.features) < 2e-2 and rel_err(logits, c.logits) < 2e-2
    assert n_probs == len(c.probs) and all(rel_err(a, b) < 2e-2 for a, b in zip(seen[:n_probs], c.probs))
