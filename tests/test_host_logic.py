"""CPU-side checks: mask sampler bit-exactness, module/state_dict layout against the reference
fixtures, C-ABI export list, loud failure on CPU tensors."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, Golden, VIT_CASES

import visiontransformer_amd as vta
from visiontransformer_amd import ViT, ViT_mask, _lib
from visiontransformer_amd.masking_generator import MaskingGenerator, seq_ord_from_unk_mask


def build_model(g):
    import visiontransformer_amd
    return getattr(visiontransformer_amd, g.module).VIT(**g.ctor_kwargs())


def test_mask_sampler_bit_exact_against_reference_vectors():
    z = np.load(os.path.join(GOLDEN, "masks.npz"), allow_pickle=False)
    count = int(z["count"])
    i = 0
    while i < count:
        grid, n, seed, rep = (int(v) for v in z[f"{i}/meta"])
        random.seed(seed)
        np.random.seed(seed)
        gen = MaskingGenerator((grid, grid), n, min_num_patches=4)
        for r in range(2):
            flat, nz, sh = gen.masked_indexes()
            assert flat.dtype == np.int32
            np.testing.assert_array_equal(flat, z[f"{i + r}/flat"])
            np.testing.assert_array_equal(sh, z[f"{i + r}/shuffle"])
            um = MaskingGenerator.unk_mask_from_flat(flat)
            assert um.dtype == np.int32 and um.shape == (1, grid * grid + 1)
            np.testing.assert_array_equal(um, z[f"{i + r}/unk_mask"])
            np.testing.assert_array_equal(seq_ord_from_unk_mask(um), z[f"{i + r}/seq_ord"])
        i += 2


def test_mask_sampler_edge_cases():
    random.seed(3)
    assert MaskingGenerator(8, 0)().sum() == 0                     # nothing requested
    m = MaskingGenerator((8, 8), 64, min_num_patches=4)()          # everything requested
    assert m.shape == (8, 8) and m.dtype == np.int32 and 0 < m.sum() <= 64
    g = MaskingGenerator(14, 98)
    assert g.min_num_patches == 8 and g.max_num_patches == 98     # reference defaults
    assert "Generator(14, 14 -> [8 ~ 98], max = 98" in repr(g)


@pytest.mark.parametrize("case", VIT_CASES)
def test_state_dict_layout_matches_reference(case):
    g = Golden(case)
    model = build_model(g)
    sd = model.state_dict()
    assert list(sd.keys()) == list(g.params.keys())
    for k, v in g.params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        assert sd[k].dtype == torch.float32
    model.load_state_dict(g.params, strict=True)                   # the reference's tensors load strictly
    for k, v in model.state_dict().items():
        assert torch.equal(v, g.params[k])


def test_constructor_contract():
    with pytest.raises(AssertionError):
        ViT.VIT(30, 4, 32, 10, 1, 2, 64)                           # image not divisible by patch
    with pytest.raises(AssertionError):
        ViT.VIT(32, 4, 32, 10, 1, 2, 64, pool="max")
    m = ViT.VIT(32, 4, 32, 10, 1, 2, 64, PE=False)
    assert m.PE is True                                            # ctor argument ignored, as in the reference
    mm = ViT_mask.VIT(32, 4, 32, 10, 1, 2, 64)
    assert mm.train_method == "zero" and mm.unk_embed_index == 65
    assert tuple(mm.pos_embedding.shape) == (1, 66, 32) and tuple(m.pos_embedding.shape) == (65, 32)
    ident = ViT.VIT(32, 8, 32, 5, 1, 1, 64, dim_head=32)
    assert isinstance(ident.transformer.layers[0][0].fn.to_out, torch.nn.Identity)
    # timm-style aliases used by trainer.py / vit_rollout.py
    assert len(m.blocks) == 1 and m.blocks[0].attn.num_heads == 2 and m.embed_dim == 32
    for blk in m.blocks:
        blk.attn.fused_attn = False
    assert m.transformer.layers[0][0].fn.fused_attn is False
    assert any("attn_drop" in n for n, _ in m.named_modules())


def test_creat_vit_reads_config_patch():
    class NS:
        pass
    cfg = NS()
    cfg.patch = NS()
    for k, v in dict(image_size=32, patch_size=4, num_classes=10, embed_dim=48, depth=1, heads=3, mlp_dim=64,
                     dropout=0, emb_dropout=0, channels=3).items():
        setattr(cfg.patch, k, v)
    m = ViT.creat_VIT(cfg)
    assert m.transformer.layers[0][0].fn.to_qkv.weight.shape == (3 * 3 * 8, 48)   # dim_head stays 8
    m2 = ViT_mask.creat_VIT(cfg)
    assert tuple(m2.pos_embedding.shape) == (1, 66, 48)


def test_cpu_tensors_fail_loudly():
    m = ViT.VIT(32, 4, 32, 10, 1, 2, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 32, 32))
    with pytest.raises(AssertionError, match="doesn't match model"):
        m(torch.randn(1, 3, 16, 16))


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vitmi.h")).read()
    declared = set(re.findall(r"\b(vitmi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vitmi.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.vitmi_version() == 100


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    # argument validation happens before any HIP call
    rc = _lib.lib.vitmi_gemm(1, 0, 4, 4, None, 1, 1, None, 1, 1, None, 4, 1, 0, None, None, 0, None, 0, None)
    assert rc == -1 and b"null operand" in _lib.lib.vitmi_last_error()
    rc = _lib.lib.vitmi_attn_fwd(1, 1, 1, None, 1, 1, 700, 1, 64, 0, None)
    assert rc == -3 and b"608" in _lib.lib.vitmi_last_error()
    with pytest.raises(_lib.VitmiError):
        _lib.check(rc, "attn")


def test_timm_style_state_dict_layout_and_rejected_switches():
    """Key names and shapes of the vendored timm class (tim/models/vision_transformer.py:440-520) and the
    switches the HIP path does not implement."""
    from visiontransformer_amd.vision_transformer import VisionTransformer, vit_small_patch16_224
    m = VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=48, depth=2, num_heads=3)
    sd = m.state_dict()
    expect = {"cls_token": (1, 1, 48), "pos_embed": (1, 66, 48), "patch_embed.proj.weight": (48, 3, 4, 4),
              "patch_embed.proj.bias": (48,), "norm.weight": (48,), "norm.bias": (48,), "head.weight": (10, 48),
              "head.bias": (10,)}
    for i in range(2):
        b = f"blocks.{i}."
        expect.update({b + "norm1.weight": (48,), b + "norm1.bias": (48,), b + "attn.qkv.weight": (144, 48),
                       b + "attn.qkv.bias": (144,), b + "attn.proj.weight": (48, 48), b + "attn.proj.bias": (48,),
                       b + "norm2.weight": (48,), b + "norm2.bias": (48,), b + "mlp.fc1.weight": (192, 48),
                       b + "mlp.fc1.bias": (192,), b + "mlp.fc2.weight": (48, 192), b + "mlp.fc2.bias": (48,)})
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect
    assert m.unk_embed_index == 65 and m.blocks[0].norm1.eps == 1e-6
    s = vit_small_patch16_224(num_classes=100)
    assert s.embed_dim == 384 and len(s.blocks) == 12 and s.blocks[0].attn.num_heads == 6
    assert tuple(s.pos_embed.shape) == (1, 198, 384)
    for bad in (dict(qk_norm=True), dict(init_values=1e-5), dict(drop_path_rate=0.1), dict(class_token=False,
                global_pool="avg"), dict(pre_norm=True)):
        with pytest.raises(NotImplementedError):
            VisionTransformer(img_size=32, patch_size=4, embed_dim=48, depth=1, num_heads=3, **bad)
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm|device"):
        m(torch.zeros(1, 3, 32, 32))


def test_presets_build_the_reference_geometries(tmp_path):
    from visiontransformer_amd import presets
    assert set(presets.list_presets()) >= {"cifar10", "cifar100", "ImageNet100", "cifar10_32x4"}
    cfg = presets.load_preset("cifar10_32x4", depth=2)
    assert cfg.patch.embed_dim == 192 and cfg["patch"]["depth"] == 2 and cfg.patch.heads == 12
    m = ViT_mask.creat_VIT(cfg)
    assert tuple(m.pos_embedding.shape) == (1, 66, 192) and len(m.transformer.layers) == 2
    assert m.transformer.layers[0][0].fn.to_qkv.weight.shape == (3 * 12 * 8, 192)      # dim_head stays 8
    assert presets.load_preset("ImageNet100").patch.depth == 8 and presets.load_preset("cifar100").patch.emb_dropout == 0.1
    with pytest.raises(KeyError, match="available"):
        presets.load_preset("nope")
    # a directory in the reference's layout: default.yaml first, the others merged over it block by block
    (tmp_path / "default.yaml").write_text("patch:\n  depth: 3\n  heads: 2\ngeneral:\n  seed: 1\n")
    (tmp_path / "config.yaml").write_text("patch:\n  depth: 5\n")
    merged = presets.load_config(str(tmp_path))
    assert merged.patch.depth == 5 and merged.patch.heads == 2 and merged.general.seed == 1
    merged.set_subkey("patch", "heads", 4)
    assert merged.values_()["patch"]["heads"] == 4


def test_the_library_is_loaded_after_torch():
    """A process that imports this package before torch must still end up with ONE HIP runtime: _lib imports torch
    (which ships its own copy of the runtime) before it loads libvitmi_hip.so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import visiontransformer_amd._lib as l; "
            "assert 'torch' in sys.modules; print('ok', l.lib.vitmi_version())" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout, r.stderr[-1500:])


def test_workspace_sizes_do_not_depend_on_the_cu_reservation():
    """Callers cache the workspace size of a shape; vitmi_set_reserved_cus (a GradReducer attached or removed) and the
    debug switches change how the weight-gradient launches split their tokens, so a size answered under one setting
    must hold under every other: the queries plan with the whole chip."""
    lib = _lib.lib
    shapes_ff = [(50432, 768, 3072), (50432, 384, 1536), (16640, 192, 512), (1970, 768, 3072)]
    shapes_attn = [(256, 197, 768, 12, 64), (256, 197, 384, 6, 64), (256, 65, 192, 12, 8)]

    def sizes():
        out = [lib.vitmi_ff_block_bwd_workspace_bytes(1, *s) for s in shapes_ff]
        out += [lib.vitmi_attn_block_bwd_workspace_bytes(1, *s) for s in shapes_attn]
        out += [lib.vitmi_gemm_wgrad_workspace_bytes(1, 768, 768, 50432), lib.vitmi_gemm_wgrad_workspace_bytes(1, 384, 1536, 50432),
                lib.vitmi_gemm_wgrad_pair_workspace_bytes(1, 50432, 768, 3072, 3072, 768),
                lib.vitmi_patch_embed_bwd_w_workspace_bytes(1, 256, 3, 224, 224, 16, 768)]
        return out

    base = sizes()
    assert all(b > 0 for b in base)
    try:
        for k in (8, 16, 32, 64, 128):
            assert lib.vitmi_set_reserved_cus(k) == 0
            assert sizes() == base, k
        lib.vitmi_set_reserved_cus(16)
        for knob, val in (("vitmi_debug_set_tn_batch", 0), ("vitmi_debug_set_tn_config", 0), ("vitmi_debug_set_tn_config", 1)):
            getattr(lib, knob)(val)
            assert sizes() == base, (knob, val)
            getattr(lib, knob)(1 if knob.endswith("batch") else -1)
    finally:
        lib.vitmi_set_reserved_cus(0)
        lib.vitmi_debug_set_tn_batch(1)
        lib.vitmi_debug_set_tn_config(-1)


@pytest.mark.parametrize("modname", ["ViT", "ViT_mask"])
def test_load_vit_round_trip(tmp_path, modname):
    """creat_VIT -> torch.save(state_dict) -> load_VIT (mymodel/ViT.py:196-211, ViT_mask.py:209-223): the reloaded
    model holds the saved tensors bit for bit, and a checkpoint of the other class is refused (their positional
    tables differ in shape)."""
    from visiontransformer_amd.presets import load_preset
    mod = getattr(vta, modname)
    cfg = load_preset("cifar10_32x4", depth=2)
    torch.manual_seed(5)
    model = mod.creat_VIT(cfg)
    path = str(tmp_path / f"{modname}_0.500.pth")
    torch.save(model.state_dict(), path)                       # what trainer.py:213-218 writes
    loaded = mod.load_VIT(path, cfg)
    assert type(loaded) is type(model)
    a, b = model.state_dict(), loaded.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    other = vta.ViT_mask if modname == "ViT" else vta.ViT
    with pytest.raises(RuntimeError, match="pos_embedding"):
        other.load_VIT(path, cfg)


def test_pbf_jigsaw_needs_the_bank_as_data():
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    with pytest.raises(ValueError, match="pub_data"):
        JigsawPuzzleMaskedRegion(32, 4, 32, 4, mask_type="pbf", device="cpu")
