"""Pins the CPU oracle (oracle/*.py) to vectors captured from the reference itself
(tests/golden/*.npz, produced by tools/gen_golden.py importing /root/reference)."""
import os
import random

import numpy as np
import torch

from conftest import GOLDEN, rel_err
from oracle import masking_oracle, vit_oracle

FP32_TOL = 2e-5   # oracle vs reference on the same CPU: only summation order differs


def test_forward_logits_features_probs(golden):
    g = golden
    probs = []
    kw = dict(heads=g.cfg["heads"], pool=g.cfg.get("pool", "cls"), unk_mask=g.unk_mask, collect_probs=probs)
    logits = vit_oracle.forward(g.params, g.x, **kw)
    assert rel_err(logits, g.logits) < FP32_TOL
    feat = vit_oracle.forward_features(g.params, g.x, g.cfg["heads"], unk_mask=g.unk_mask)
    assert rel_err(feat, g.feat) < FP32_TOL
    assert len(probs) == len(g.probs)
    for i, p in enumerate(probs):
        assert p.shape == g.probs[i].shape
        assert rel_err(p, g.probs[i]) < FP32_TOL


def test_loss_and_grads(golden):
    g = golden
    logits, loss, grads = vit_oracle.loss_and_grads(
        g.params, g.x, g.y, g.cfg["heads"], pool=g.cfg.get("pool", "cls"), unk_mask=g.unk_mask)
    assert abs(float(loss) - float(g.loss)) < 1e-5 * max(1.0, abs(float(g.loss)))
    assert set(grads) == set(g.grads)
    for k, ref in g.grads.items():
        assert grads[k].shape == ref.shape, k
        assert rel_err(grads[k], ref) < 5e-5, k


def test_masked_pe_unknown_row_collects_gradient():
    from conftest import Golden
    g = Golden("mask_dh8")
    n = g.unk_mask.shape[1]
    gpe = g.grads["pos_embedding"]
    assert gpe.shape == (1, n + 1, 96)
    masked = g.unk_mask[0].bool()
    assert gpe[0, :n][masked].abs().max() == 0        # masked positions never read their own row
    assert gpe[0, n].abs().max() > 0                  # the unknown slot did
    g2 = Golden("mask_none")
    assert g2.grads["pos_embedding"][0, n].abs().max() == 0


def test_mask_vectors_bit_exact():
    z = np.load(os.path.join(GOLDEN, "masks.npz"), allow_pickle=False)
    count = int(z["count"])
    assert count >= 80
    i = 0
    while i < count:
        grid, n, seed, rep = (int(v) for v in z[f"{i}/meta"])
        assert rep == 0
        random.seed(seed)
        np.random.seed(seed)
        for r in range(2):
            flat, nz, sh = masking_oracle.masked_indexes(grid, grid, n, 4)
            assert flat.dtype == np.int32
            np.testing.assert_array_equal(flat, z[f"{i + r}/flat"])
            np.testing.assert_array_equal(sh, z[f"{i + r}/shuffle"])
            um = masking_oracle.unk_mask_from_flat(flat)
            np.testing.assert_array_equal(um, z[f"{i + r}/unk_mask"])
            np.testing.assert_array_equal(masking_oracle.seq_ord(um), z[f"{i + r}/seq_ord"])
        i += 2


def test_mask_can_fall_short_of_request():
    # the sampler gives up after ten fruitless attempts: (14x14, 98) under seed 1234 of
    # the reference's own generator yields fewer than 98 -- the fixture records what it gave
    z = np.load(os.path.join(GOLDEN, "masks.npz"), allow_pickle=False)
    short = 0
    for i in range(int(z["count"])):
        grid, n, seed, rep = (int(v) for v in z[f"{i}/meta"])
        s = int(z[f"{i}/flat"].sum())
        assert s <= n
        short += s < n
    assert short > 0
