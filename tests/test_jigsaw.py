"""JigsawPuzzleMaskedRegion: the numpy oracle against vectors captured from the reference -- "mjp" from the
importable utils/masking_generator (jigsaw_mjp.npz), every other branch from the class definitions of the top-level
masking_generator.py compiled out of the file (its imports need torchvision), with the public patch bank handed in as
data (jigsaw_modes.npz; tools/gen_golden.py) -- and the HIP gather against both.  Patch movement is pure copying, so
those branches are compared bit for bit; "repeat" averages in fp32 (summation order differs: 2e-6 / one bf16 ulp)."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import jigsaw_oracle

DEV = "cuda:0"


def golden():
    z = np.load(os.path.join(GOLDEN, "jigsaw_mjp.npz"), allow_pickle=False)
    for k in z["keys"]:
        img, p, nmask, nmin, B, C, seed = (int(v) for v in z[str(k) + "/meta"])
        yield str(k), dict(img=img, p=p, nmask=nmask, nmin=nmin, B=B, C=C, seed=seed), z[str(k) + "/x"], \
            z[str(k) + "/out"], z[str(k) + "/masks"]


def test_oracle_matches_reference_vectors():
    n = 0
    for k, m, x, out, masks in golden():
        random.seed(m["seed"]), np.random.seed(m["seed"])
        o, mk = jigsaw_oracle.jigsaw(x.copy(), m["p"], m["nmask"], m["nmin"], "mjp")
        np.testing.assert_array_equal(o, out, err_msg=k)
        np.testing.assert_array_equal(mk, masks, err_msg=k)
        assert mk.dtype == np.int32 and mk.shape == (1, (m["img"] // m["p"]) ** 2 + 1)
        n += 1
    assert n == 7


def golden_modes():
    z = np.load(os.path.join(GOLDEN, "jigsaw_modes.npz"), allow_pickle=False)
    for k in z["keys"]:
        k = str(k)
        g, mode, seed = k.split("/")
        img, p, nmask, nmin, B, C = (int(v) for v in z[g + "/meta"])
        masks = z[k + "/masks"]
        yield k, dict(img=img, p=p, nmask=nmask, nmin=nmin, B=B, C=C, seed=int(seed), mode=mode), z[k + "/x"], \
            z[k + "/out"], (masks if masks.size else None), z[g + "/bank"]


def test_oracle_matches_reference_vectors_of_every_branch():
    seen = set()
    for k, m, x, out, masks, bank in golden_modes():
        random.seed(m["seed"]), np.random.seed(m["seed"]), torch.manual_seed(m["seed"])
        o, mk = jigsaw_oracle.jigsaw(x.copy(), m["p"], m["nmask"], m["nmin"], m["mode"], bank)
        if m["mode"] == "repeat":
            np.testing.assert_allclose(o, out, rtol=0, atol=2e-6 * max(1.0, np.abs(out).max()), err_msg=k)
        else:
            np.testing.assert_array_equal(o, out, err_msg=k)
        if masks is None:
            assert mk is None, k
        else:
            np.testing.assert_array_equal(mk, masks, err_msg=k)
        seen.add(m["mode"])
    assert seen == {"pbf", "orain", "zero", "repeat", "exchange", "Scale", "noise", "pps"}


def test_oracle_unfold_fold_round_trip_and_branch_properties():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 12, 12)).astype(np.float32)
    cols = jigsaw_oracle.unfold(x, 4)
    assert cols.shape == (2, 48, 9)
    np.testing.assert_array_equal(cols, torch.nn.Unfold((4, 4), stride=4)(torch.from_numpy(x)).numpy())
    np.testing.assert_array_equal(jigsaw_oracle.fold(cols, 3, 12, 12, 4), x)
    flat = np.array([0, 1, 1, 0, 0, 1, 0, 0, 0], dtype=np.int32)
    nz, sh = np.array([1, 2, 5]), np.array([5, 1, 2])
    z = jigsaw_oracle.apply(x, 4, "zero", flat, nz, sh)
    assert np.all(z[:, :, 0:4, 4:12] == 0) and np.array_equal(z[:, :, 8:, :], x[:, :, 8:, :])
    pub = rng.standard_normal((5, 48, 9)).astype(np.float32)
    b = jigsaw_oracle.apply(x, 4, "pbf", flat, nz, sh, pub, [3, 0])
    np.testing.assert_array_equal(jigsaw_oracle.unfold(b, 4)[1][:, 1], pub[0][:, 5])
    r = jigsaw_oracle.apply(x, 4, "repeat", flat, nz, sh)
    want = (cols[:, :, [0, 3, 4, 6, 7, 8]].sum(2) + 3 * cols[:, :, 0]) / 9
    np.testing.assert_allclose(jigsaw_oracle.unfold(r, 4)[:, :, 2], want, rtol=1e-5, atol=1e-6)
    assert jigsaw_oracle.apply(x, 4, "orain", flat, nz, sh) is x


def test_host_class_contract_without_a_gpu():
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    with pytest.raises(ValueError, match="pbf"):
        JigsawPuzzleMaskedRegion(32, 4, 8, 2, mask_type="pbf")
    x0 = torch.zeros(1, 3, 32, 32)
    state = (random.getstate(), np.random.get_state()[1].copy())
    out, masks = JigsawPuzzleMaskedRegion(32, 4, 8, 2, mask_type="pps", device="cpu")(x0)   # the reference's final else
    assert out is x0 and masks is None
    assert random.getstate() == state[0] and np.array_equal(np.random.get_state()[1], state[1])   # no mask drawn
    # the bank is handed in by the caller ([P, C*p*p, L], the tensor dataloader.py:17-37 builds); no module is imported
    bank = torch.arange(5 * 48 * 64, dtype=torch.float32).view(5, 48, 64)
    jp = JigsawPuzzleMaskedRegion(32, 4, 8, 2, mask_type="pbf", pub_data_dir="/data/public/", device="cpu", pub_data=bank)
    assert tuple(jp.pub_data.shape) == (5, 48, 64) and tuple(jp._bank.shape) == (5, 3, 32, 32)
    j = JigsawPuzzleMaskedRegion(32, 4, 16, 2, mask_type="orain", device="cpu")
    random.seed(3), np.random.seed(3)
    x = torch.zeros(2, 3, 32, 32)
    out, masks = j(x)
    assert out is x and masks.shape == (1, 65) and masks.dtype == np.int32 and masks[0, 0] == 0
    j._update_masking_generator(10, 1, masking_ratio=0.25)
    assert j.masking_generator.num_masking_patches == 16
    jm = JigsawPuzzleMaskedRegion(32, 4, 16, 2, mask_type="mjp", device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        jm(x)
    with pytest.raises(ValueError, match="pub_data must be"):
        JigsawPuzzleMaskedRegion(32, 4, 8, 2, mask_type="pbf", device="cpu", pub_data=torch.zeros(4, 48, 63))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_mjp_matches_reference_vectors(dtype):
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    for k, m, x, out, masks in golden():
        j = JigsawPuzzleMaskedRegion(m["img"], m["p"], m["nmask"], m["nmin"], mask_type="mjp", channels=m["C"], device=DEV)
        random.seed(m["seed"]), np.random.seed(m["seed"])
        xt = torch.from_numpy(x).to(DEV).to(dtype)
        keep = xt.clone()
        got, mk = j(xt)
        np.testing.assert_array_equal(mk, masks, err_msg=k)
        assert torch.equal(xt, keep)                                    # the input batch is left as it was
        want = torch.from_numpy(out).to(DEV).to(dtype)                  # a copy commutes with the rounding
        assert got.dtype == dtype and torch.equal(got, want), k


@pytest.mark.gpu
def test_hip_branches_match_reference_vectors():
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    n = 0
    for k, m, x, out, masks, bank in golden_modes():
        j = JigsawPuzzleMaskedRegion(m["img"], m["p"], m["nmask"], m["nmin"], mask_type=m["mode"], channels=m["C"],
                                     device=DEV, pub_data=torch.from_numpy(bank))
        random.seed(m["seed"]), np.random.seed(m["seed"]), torch.manual_seed(m["seed"])
        got, mk = j(torch.from_numpy(x).to(DEV))
        if masks is None:
            assert mk is None, k
        else:
            np.testing.assert_array_equal(mk, masks, err_msg=k)
        if m["mode"] == "repeat":
            assert (got.cpu() - torch.from_numpy(out)).abs().max().item() <= 2e-6 * max(1.0, np.abs(out).max()), k
        else:
            assert torch.equal(got.cpu(), torch.from_numpy(out)), k
        n += 1
    assert n == 48


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["mjp", "pbf", "zero", "repeat", "orain", "exchange", "Scale", "noise"])
@pytest.mark.parametrize("geom", [(32, 4, 32, 3, 5), (224, 16, 98, 3, 4), (30, 3, 40, 1, 2), (28, 14, 2, 2, 3)])
def test_hip_branches_against_oracle(geom, mode, dtype):
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    img, p, nmask, C, B = geom
    L = (img // p) ** 2
    g = torch.Generator().manual_seed(img * 31 + p)
    x = torch.randn(B, C, img, img, generator=g).to(dtype)
    pub = torch.randn(7, C * p * p, L, generator=g).to(dtype)
    j = JigsawPuzzleMaskedRegion(img, p, nmask, 1, mask_type=mode, channels=C, device=DEV, pub_data=pub)
    for seed in (11, 12):
        random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
        got, mk = j(x.to(DEV))
        random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
        want, wmk = jigsaw_oracle.jigsaw(x.float().numpy().copy(), p, nmask, 1, mode, pub.float().numpy())
        if wmk is None:
            assert mk is None
        else:
            np.testing.assert_array_equal(mk, wmk)
        want = torch.from_numpy(want)
        if mode in ("Scale", "noise"):
            # one multiply / one add in fp32: exact in fp32 mode, one rounding to bf16 otherwise
            if dtype == torch.float32:
                assert torch.equal(got.cpu(), want)
            else:
                assert torch.equal(got.cpu(), want.to(torch.bfloat16))
        elif mode == "repeat":
            tol = 1e-6 if dtype == torch.float32 else 2 ** -7
            err = (got.float().cpu() - want).abs().max().item()
            assert err <= tol * max(1.0, want.abs().max().item()), err
        else:
            assert torch.equal(got.float().cpu(), want)                 # copies: bit for bit


@pytest.mark.gpu
def test_hip_gather_full_size_properties():
    """ImageNet geometry at the training batch: "mjp" permutes patches, so per-image multisets of pixel values are
    kept and applying the inverse table restores the batch; "zero" clears exactly the masked area."""
    from visiontransformer_amd import ops
    B, C, img, p = 256, 3, 224, 16
    L = (img // p) ** 2
    x = torch.randn(B, C, img, img, device=DEV)
    perm = torch.randperm(L)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(L)
    y = ops.jigsaw_gather(x, p, perm.to(torch.int32).to(DEV))
    assert not torch.equal(x, y)
    assert torch.equal(ops.jigsaw_gather(y, p, inv.to(torch.int32).to(DEV)), x)
    assert torch.equal(y.sum(dtype=torch.float64), x.sum(dtype=torch.float64)) or \
        abs(float(y.double().sum() - x.double().sum())) < 1e-6
    src = torch.arange(L, dtype=torch.int32)
    src[::3] = -1
    z = ops.jigsaw_gather(x, p, src.to(DEV))
    zp = z.view(B, C, img // p, p, img // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, L, -1)
    xp = x.view(B, C, img // p, p, img // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, L, -1)
    assert torch.all(zp[:, ::3] == 0) and torch.equal(zp[:, 1::3], xp[:, 1::3]) and torch.equal(zp[:, 2::3], xp[:, 2::3])
    with pytest.raises(ValueError):
        ops.jigsaw_gather(x, p, src[:-1].to(DEV))
