"""Attention rollout: the torch-CPU oracle against vectors produced by the reference's own function bodies
(vit_rollout.rollout / last_attn, trainer.rollout), and the HIP path (vitmi_rollout) against both.
Bar: fp32 1e-4 relative (max|a-b| / max|b|); the HIP path carries row 0 through a vector-matrix chain where
the reference multiplies whole matrices, so the fp32 summation order differs.  bf16 probabilities: 2e-2."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import rollout_oracle

DEV = "cuda:0"


def cases():
    z = np.load(os.path.join(GOLDEN, "rollout.npz"), allow_pickle=False)
    for k in z["keys"]:
        k = str(k)
        c, fusion, ratio = k.split("/")
        B, H, N, L = (int(v) for v in z[c + "/meta"])
        atts = [torch.from_numpy(z[f"{c}/att{l}"]) for l in range(L)]
        yield k, atts, fusion, float(ratio), {n: torch.from_numpy(z[f"{k}/{n}"]) for n in ("roll", "trainer", "last")}


def test_oracle_matches_reference_vectors():
    n = 0
    for k, atts, fusion, ratio, want in cases():
        keep = [a.clone() for a in atts]
        # bit-identical on the machine that captured the vectors; torch picks its matmul kernel by CPU, so on another
        # host the chain of [N,N] products may round differently in the last bit
        assert rel_err(rollout_oracle.rollout(atts, ratio, fusion), want["roll"]) < 1e-6, k
        assert rel_err(rollout_oracle.trainer_rollout(atts, ratio, fusion), want["trainer"]) < 1e-6, k
        assert torch.equal(rollout_oracle.last_attn(atts, fusion), want["last"]), k
        assert all(torch.equal(a, b) for a, b in zip(atts, keep))            # inputs untouched
        n += 1
    assert n == 19


def test_oracle_properties():
    torch.manual_seed(0)
    atts = [torch.softmax(torch.randn(2, 3, 9, 9), -1) for _ in range(3)]
    r = rollout_oracle.rollout(atts, 0.5, "mean")
    assert r.shape == (2, 8) and torch.allclose(r.sum(-1), torch.ones(2), atol=1e-6)
    t = rollout_oracle.trainer_rollout(atts, 0.0, "mean")
    a = [rollout_oracle.normalised_layer(x, 0.0, "mean", True) for x in atts]
    assert torch.allclose(t, (a[2] @ a[1] @ a[0])[:, 0, 1:], atol=1e-7)
    full = rollout_oracle.normalised_layer(atts[0], 1.0, "mean", False)       # everything discarded -> identity
    assert torch.allclose(full, torch.eye(9).expand(2, 9, 9))


def test_host_module_rejects_bad_input_without_a_gpu():
    from visiontransformer_amd import vit_rollout
    a = torch.softmax(torch.randn(1, 2, 5, 5), -1)
    with pytest.raises(ValueError, match="fusion"):
        vit_rollout.rollout([a], 0.9, "median")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vit_rollout.rollout([a], 0.9, "mean")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vit_rollout.last_attn([a], "max")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_rollout_matches_reference_vectors(dtype):
    from visiontransformer_amd import vit_rollout
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    for k, atts, fusion, ratio, want in cases():
        dev = [a.to(DEV).to(dtype) for a in atts]
        if dtype == torch.bfloat16:       # the rounded probabilities are the input: restate the target on them
            rounded = [a.float().cpu() for a in dev]
            want = {"roll": rollout_oracle.rollout(rounded, ratio, fusion),
                    "trainer": rollout_oracle.trainer_rollout(rounded, ratio, fusion),
                    "last": rollout_oracle.last_attn(rounded, fusion)}
            if fusion != "mean" and ratio not in (0.0, 1.0):
                continue                   # max/min of rounded values tie at the threshold, where torch.topk's choice
                                           # is unspecified; the tie rule has its own test below
        got = vit_rollout.rollout(dev, ratio, fusion, DEV)
        assert got.dtype == torch.float32 and rel_err(got.cpu(), want["roll"]) < tol, (k, "roll")
        got = vit_rollout.trainer_rollout(dev, ratio, fusion)
        assert rel_err(got.cpu(), want["trainer"]) < tol, (k, "trainer")
        got = vit_rollout.last_attn(dev, fusion)
        assert rel_err(got.cpu(), want["last"]) < (1e-6 if dtype == torch.float32 else tol), (k, "last")


def _discard_lowest_index_first(fused, k, first):
    """numpy statement of the documented tie rule: the k smallest by (value, flat index)."""
    out = fused.copy()
    for b in range(out.shape[0]):
        flat = out[b].reshape(-1)
        cand = np.arange(first, flat.size)
        order = cand[np.lexsort((cand, flat[cand]))]
        flat[order[:min(k, cand.size)]] = 0
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("keep_first", [False, True])
def test_hip_discard_with_ties_takes_lowest_indices_first(keep_first):
    """Quantised probabilities (many equal values): single layer, raw row, against the tie rule in numpy."""
    from visiontransformer_amd import vit_rollout
    torch.manual_seed(3)
    B, H, N = 3, 2, 33
    p = torch.softmax(torch.randn(B, H, N, N), -1)
    p = (p * 64).round() / 64                                  # a handful of distinct values
    for ratio in (0.3, 0.9, 1.0):
        k = int(N * N * ratio)
        fused = p.mean(1).numpy()
        kept = _discard_lowest_index_first(fused, k, int(keep_first))
        a = (kept + np.eye(N, dtype=np.float32)) / 2
        a = a / a.sum(-1, keepdims=True)
        got = vit_rollout._rollout([p.to(DEV)], ratio, "mean", keep_first=keep_first, final_mode=0).cpu().numpy()
        np.testing.assert_allclose(got, a[:, 0, 1:], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_hip_rollout_full_size_against_oracle_and_properties():
    """ViT-B/16 geometry (12 layers x [B,12,197,197]); oracle on a 4-sample slice, sum-to-one on the whole batch."""
    from visiontransformer_amd import vit_rollout
    torch.manual_seed(11)
    B, H, N, L = 32, 12, 197, 12
    atts = [torch.softmax(1.5 * torch.randn(B, H, N, N, device=DEV), -1) for _ in range(L)]
    got = vit_rollout.rollout(atts, 0.9, "mean", DEV)
    assert got.shape == (B, N - 1) and torch.allclose(got.sum(-1), torch.ones(B, device=DEV), atol=1e-5)
    raw = vit_rollout.trainer_rollout(atts, 0.9, "mean")
    assert torch.all(raw >= 0) and torch.all(raw.sum(-1) < 1.0 + 1e-5)       # rows of a stochastic matrix product
    small = [a[:4].cpu() for a in atts]
    assert rel_err(got[:4].cpu(), rollout_oracle.rollout(small, 0.9, "mean")) < 1e-4
    assert rel_err(raw[:4].cpu(), rollout_oracle.trainer_rollout(small, 0.9, "mean")) < 1e-4


@pytest.mark.gpu
def test_rollout_class_on_the_model():
    """VITAttentionRollout hooked on the drop-in model (vit_rollout.py:77-105) against the oracle fed with the same
    exported probabilities."""
    from visiontransformer_amd import ViT, vit_rollout
    torch.manual_seed(5)
    m = ViT.VIT(32, 4, 64, 10, 3, 4, 128, dim_head=16).to(DEV).set_compute_dtype(torch.float32).eval()
    for blk in m.blocks:
        blk.attn.fused_attn = False                                        # probability export on
    ro = vit_rollout.VITAttentionRollout(m, head_fusion="max", discard_ratio=0.8, device=DEV)
    x = torch.randn(3, 3, 32, 32, device=DEV)
    roll = ro(x, "roll")
    assert len(ro.attentions) == 3 and tuple(ro.attentions[0].shape) == (3, 4, 65, 65)
    want = rollout_oracle.rollout([a.float().cpu() for a in ro.attentions], 0.8, "max")
    assert rel_err(roll.cpu(), want) < 1e-4
    la = ro(x, "attn")
    assert rel_err(la.cpu(), rollout_oracle.last_attn([a.float().cpu() for a in ro.attentions], "max")) < 1e-6
    out = ro(x, "out")
    assert tuple(out.shape) == (3, 64)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [203, 256])
def test_hip_rollout_beyond_the_register_path(N):
    """Token counts whose fused matrix does not fit the per-thread registers take the in-memory selection path."""
    from visiontransformer_amd import vit_rollout
    torch.manual_seed(N)
    atts = [torch.softmax(2.0 * torch.randn(2, 2, N, N), -1) for _ in range(2)]
    dev = [a.to(DEV) for a in atts]
    for fusion, ratio in (("mean", 0.9), ("min", 0.4)):
        assert rel_err(vit_rollout.rollout(dev, ratio, fusion).cpu(), rollout_oracle.rollout(atts, ratio, fusion)) < 1e-4
        assert rel_err(vit_rollout.trainer_rollout(dev, ratio, fusion).cpu(),
                       rollout_oracle.trainer_rollout(atts, ratio, fusion)) < 1e-4
    q = [(a * 256).round() / 256 for a in atts[:1]]                       # ties in the in-memory path
    k = int(N * N * 0.7)
    kept = _discard_lowest_index_first(q[0].mean(1).numpy(), k, 0)
    a = (kept + np.eye(N, dtype=np.float32)) / 2
    a = a / a.sum(-1, keepdims=True)
    got = vit_rollout._rollout([q[0].to(DEV)], 0.7, "mean", keep_first=False, final_mode=0).cpu().numpy()
    np.testing.assert_allclose(got, a[:, 0, 1:], rtol=1e-5, atol=1e-7)
    with pytest.raises(Exception, match="256"):
        vit_rollout.rollout([torch.softmax(torch.randn(1, 1, 257, 257, device=DEV), -1)], 0.9, "mean")


# ------------------------------------------------------------------------------------------ gradient rollout
def grad_cases():
    z = np.load(os.path.join(GOLDEN, "grad_rollout.npz"), allow_pickle=False)
    for k in z["keys"]:
        k = str(k)
        c, ratio = k.split("/")
        H, N, L = (int(v) for v in z[c + "/meta"])
        atts = [torch.from_numpy(z[f"{c}/att{l}"]) for l in range(L)]
        grads = [torch.from_numpy(z[f"{c}/grad{l}"]) for l in range(L)]
        yield k, atts, grads, float(ratio), z[k + "/mask"]


def test_grad_rollout_oracle_matches_reference_vectors():
    """oracle/rollout_oracle.grad_rollout against the outputs of the reference's own function body
    (vit_grad_rollout.py:9-36, compiled out of the file by tools/gen_golden.py): bit for bit where the vectors were
    captured, 1e-6 on hosts whose matmul kernels differ."""
    n = 0
    for k, atts, grads, ratio, want in grad_cases():
        got = rollout_oracle.grad_rollout([a.clone() for a in atts], [g.clone() for g in grads], ratio)
        assert got.shape == want.shape and rel_err(torch.from_numpy(got), torch.from_numpy(want)) < 1e-6, k
        n += 1
    assert n == 9


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_grad_rollout_against_reference_vectors(dtype):
    """vitmi_grad_rollout against the reference's vectors (fp32 1e-4) and, for bf16 inputs, against the oracle run on
    the rounded inputs (2e-2); a batch treats every image as the reference treats its one."""
    from visiontransformer_amd import vit_rollout
    for k, atts, grads, ratio, want in grad_cases():
        a = [t.to(DEV, dtype) for t in atts]
        g = [t.to(DEV, dtype) for t in grads]
        got = vit_rollout.grad_rollout(a, g, ratio)
        if dtype == torch.float32:
            ref, tol = want, 1e-4
        else:
            ref, tol = rollout_oracle.grad_rollout([t.float().cpu() for t in a], [t.float().cpu() for t in g], ratio), 2e-2
        assert got.shape == ref.shape and rel_err(torch.from_numpy(got), torch.from_numpy(ref)) < tol, k
    # batch of two different images = the two single-image results
    k, atts, grads, ratio, want = next(grad_cases())
    a2 = [torch.cat([t, t.flip(1)]).to(DEV) for t in atts]
    g2 = [torch.cat([t, t.flip(1) * 0.5]).to(DEV) for t in grads]
    both = vit_rollout.grad_rollout(a2, g2, ratio)
    one = vit_rollout.grad_rollout([t[1:] for t in a2], [t[1:] for t in g2], ratio)
    assert both.shape == (2,) + want.shape and rel_err(torch.from_numpy(both[0]), torch.from_numpy(want)) < 1e-4
    assert np.array_equal(both[1], one)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_probability_gradients_reach_backward_hooks(dtype):
    """A backward hook on attn_drop (vit_grad_rollout.py:44-46) turns the attention block into the chain
    softmax -> tap -> probs @ v of separate autograd nodes: the hooks see dL/dP of every layer, logits and parameter
    gradients stay those of the fused block, and VITAttentionGradRollout equals the oracle's grad_rollout of the
    probabilities and gradients the CPU restatement of the model produces (mymodel/ViT.py:93-104 under autograd)."""
    from oracle import vit_oracle
    from visiontransformer_amd import ViT, vit_rollout
    kw = dict(image_size=32, patch_size=8, embed_dim=64, num_classes=10, depth=2, heads=4, mlp_dim=128, dim_head=16)
    params = vit_oracle.init_params(masked=False, seed=9, **kw)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 32, 32, generator=g)
    cat = 3
    # CPU: probabilities that require grad, captured per layer
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    probs_cpu = []
    logits = vit_oracle.forward(leaf, x, 4, collect_probs=probs_cpu)
    for p in probs_cpu:
        p.retain_grad()
    logits[0, cat].backward()
    tol = 1e-4 if dtype == torch.float32 else 2e-2

    m = ViT.VIT(**kw)
    m.load_state_dict(params)
    m = m.to(DEV).set_compute_dtype(dtype).train()
    fused_logits = m(x.to(DEV))
    fused_logits[0, cat].backward()
    fused_grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    roll = vit_rollout.VITAttentionGradRollout(m, discard_ratio=0.5, match_layers=True)
    mask = roll(x.to(DEV), cat)
    assert len(roll.attentions) == 2 and len(roll.attention_gradients) == 2
    for l in range(2):
        assert rel_err(roll.attentions[l], probs_cpu[l]) < tol
        assert rel_err(roll.attention_gradients[1 - l], probs_cpu[l].grad) < tol      # arrival order: last layer first
    for k, p in m.named_parameters():
        assert rel_err(p.grad, leaf[k].grad) < tol, k
        assert rel_err(p.grad, fused_grads[k]) < tol, k
    want = rollout_oracle.grad_rollout([p.detach() for p in probs_cpu], [p.grad for p in probs_cpu], 0.5)
    assert mask.shape == (4, 4) and rel_err(torch.from_numpy(mask), torch.from_numpy(want)) < (1e-3 if dtype == torch.float32 else 5e-2)
    # the reference's own pairing (gradients as they arrive)
    ref_pair = vit_rollout.VITAttentionGradRollout(m, discard_ratio=0.5)(x.to(DEV), cat)
    want_ref = rollout_oracle.grad_rollout([p.detach() for p in probs_cpu], [p.grad for p in probs_cpu][::-1], 0.5)
    assert rel_err(torch.from_numpy(ref_pair), torch.from_numpy(want_ref)) < (1e-3 if dtype == torch.float32 else 5e-2)
