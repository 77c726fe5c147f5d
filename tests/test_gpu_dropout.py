"""Dropout inside the blocks (mymodel/ViT.py:68,70,90 with dropout > 0, training mode).  The HIP path draws its masks
from a counter-based generator keyed by a per-site seed; the test replays the seeds, reads each site's multiplier
tensor back through the same kernel (dropout of a tensor of ones) and hands those multipliers to the CPU oracle.
Bars as everywhere: fp32 1e-4, bf16 2e-2 relative."""
import pytest
import torch

from conftest import rel_err
from oracle import vit_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def test_dropout_kernel_statistics_and_backward():
    from visiontransformer_amd import ops
    x = torch.ones(1 << 20, device=DEV)
    for p in (0.1, 0.5, 0.9):
        y = ops.dropout(x, p, 1234)
        kept = (y != 0).float().mean().item()
        assert abs(kept - (1 - p)) < 3e-3
        assert torch.allclose(y[y != 0], torch.full((1,), 1 / (1 - p), device=DEV))
        assert torch.equal(y, ops.dropout(x, p, 1234)) and not torch.equal(y, ops.dropout(x, p, 1235))
    assert torch.equal(ops.dropout(x, 0.0, 7), x)
    # odd length, bf16, in place; the mask does not depend on the dtype
    a = torch.ones(1001, device=DEV)
    b = a.bfloat16()
    ya = ops.dropout(a, 0.5, 99)
    ops.dropout(b, 0.5, 99, out=b)
    assert torch.equal(ya.bfloat16(), b)
    z = torch.randn(257, 33, device=DEV, requires_grad=True)
    out = ops.DropoutFn.apply(z, 0.3, 5)
    out.sum().backward()
    assert torch.equal(z.grad, ops.dropout(torch.ones_like(z), 0.3, 5))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_with_dropout_against_oracle(dtype, monkeypatch):
    from visiontransformer_amd import ViT, ops
    kw = dict(image_size=32, patch_size=4, embed_dim=192, num_classes=10, depth=2, heads=12, mlp_dim=512)
    p = 0.2
    params = vit_oracle.init_params(masked=False, seed=4, **kw)
    gen = torch.Generator().manual_seed(6)
    B, N, D, M = 6, 65, 192, 512
    x = torch.randn(B, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (B,), generator=gen)
    seeds = iter(range(1000, 2000))
    monkeypatch.setattr(ops, "new_seed", lambda: next(seeds))
    m = ViT.VIT(dropout=p, **kw)
    m.load_state_dict(params)
    m = m.to(DEV).set_compute_dtype(dtype).train()
    logits = m(x.to(DEV))
    # call order per layer: attention output, then (hidden, MLP output)
    keep, s = {}, 1000
    for i in range(kw["depth"]):
        for site, width in (("attn_out", D), ("hidden", M), ("ff_out", D)):
            keep[(i, site)] = ops.dropout(torch.ones(B, N, width, device=DEV), p, s).cpu()
            s += 1
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    ref = vit_oracle.forward(leaf, x, kw["heads"], keep=keep)
    assert rel_err(logits, ref.detach()) < TOL[dtype]
    vit_oracle.cross_entropy(ref, y).backward()
    torch.nn.functional.cross_entropy(logits, y.to(DEV)).backward()
    worst = max((rel_err(q.grad, leaf[k].grad), k) for k, q in m.named_parameters())
    assert worst[0] < 2 * TOL[dtype], worst
    # eval mode: no dropout, the fused path
    m.eval()
    with torch.no_grad():
        assert rel_err(m(x.to(DEV)), vit_oracle.forward(params, x, kw["heads"])) < TOL[dtype]
