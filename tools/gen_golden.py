#!/usr/bin/env python3
"""Capture golden vectors from the reference itself (runs ONLY in the build
container, where /root/reference is mounted; never on the GPU box).

Imports mymodel.ViT, mymodel.ViT_mask and utils.masking_generator from the
reference, drives them on seeded inputs with torch 2.10 CPU fp32, and writes
small .npz fixtures (plain arrays, no pickles) under tests/golden/:

  vit_<case>.npz   state_dict ("p/<key>"), x, y, logits, loss, grads ("g/<key>"),
                   per-layer post-softmax probabilities ("probs/<i>") taken with a
                   forward hook on each Attention.attend module, token features
                   before pooling ("feat"), config scalars ("cfg/<name>")
  masks.npz        MaskingGenerator outputs for several (grid, n, seed) triples,
                   with the shuffled index copy, unk_mask and seq_ord

Usage: PYTHONDONTWRITEBYTECODE=1 python3 tools/gen_golden.py
"""
import os
import random
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CASES = {
    # name: (module, ctor kwargs, batch, masked?, train_method)
    "cifar_dh8": ("ViT", dict(image_size=32, patch_size=4, embed_dim=96, num_classes=10, depth=2,
                              heads=12, mlp_dim=128), 4, None),
    "dh64_mean": ("ViT", dict(image_size=64, patch_size=16, embed_dim=128, num_classes=100, depth=2,
                              heads=2, mlp_dim=256, dim_head=64, pool="mean"), 3, None),
    "n197_dh64": ("ViT", dict(image_size=224, patch_size=16, embed_dim=64, num_classes=10, depth=1,
                              heads=2, mlp_dim=128, dim_head=64), 2, None),
    "identity_out": ("ViT", dict(image_size=32, patch_size=8, embed_dim=32, num_classes=5, depth=1,
                                 heads=1, mlp_dim=64, dim_head=32), 2, None),
    "mask_dh8": ("ViT_mask", dict(image_size=32, patch_size=4, embed_dim=96, num_classes=10, depth=2,
                                  heads=12, mlp_dim=128), 4, "mask"),
    "mask_none": ("ViT_mask", dict(image_size=32, patch_size=4, embed_dim=96, num_classes=10, depth=2,
                                   heads=12, mlp_dim=128), 4, "nomask"),
    "mask_dh64": ("ViT_mask", dict(image_size=64, patch_size=8, embed_dim=64, num_classes=7, depth=1,
                                   heads=2, mlp_dim=96, dim_head=32), 3, "mask"),
    # sibling positional-embedding variants (mymodel/ViT_mask_avg.py, mymodel/ViT_ape.py)
    "avg_mask": ("ViT_mask_avg", dict(image_size=32, patch_size=4, embed_dim=96, num_classes=10, depth=2,
                                      heads=12, mlp_dim=128), 4, "mask"),
    "avg_none": ("ViT_mask_avg", dict(image_size=32, patch_size=4, embed_dim=96, num_classes=10, depth=2,
                                      heads=12, mlp_dim=128), 4, "nomask"),
    # the masked branch of ViT_ape reads an undefined attribute (ViT_ape.py:186): only unmasked is captured
    "ape_none": ("ViT_ape", dict(image_size=32, patch_size=4, embed_dim=96, num_classes=10, depth=2,
                                 heads=12, mlp_dim=128), 4, "nomask"),
}


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from mymodel import ViT, ViT_ape, ViT_mask, ViT_mask_avg
    from utils.masking_generator import MaskingGenerator
    modules = {"ViT": ViT, "ViT_mask": ViT_mask, "ViT_mask_avg": ViT_mask_avg, "ViT_ape": ViT_ape}
    only = set(sys.argv[1:])          # optional: case names to (re)generate; masks.npz only on a full run

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # deterministic reduction order for the captured numbers

    for name, (modname, kw, B, mode) in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(1001)
        mod = modules[modname]
        model = mod.VIT(**kw)
        model.train()
        # make LayerNorm affine / biases non-trivial so their gradients and use are exercised
        with torch.no_grad():
            for k, v in model.state_dict().items():
                if k.endswith("norm.weight") or k == "mlp_head.0.weight":
                    v.add_(0.1 * torch.randn_like(v))
                if k.endswith("norm.bias") or k == "mlp_head.0.bias":
                    v.add_(0.1 * torch.randn_like(v))
        torch.manual_seed(1234)
        x = torch.randn(B, 3, kw["image_size"], kw["image_size"])
        y = torch.randint(0, kw["num_classes"], (B,))
        probs = []
        hooks = []
        for layer in model.transformer.layers:
            hooks.append(layer[0].fn.attend.register_forward_hook(lambda m, i, o: probs.append(o.detach().clone())))
        feats = []
        hooks.append(model.transformer.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone())))
        arrays = {}
        if modname != "ViT":
            if modname == "ViT_mask":
                model.train_method = "mask"
            if mode == "mask":
                grid = kw["image_size"] // kw["patch_size"]
                random.seed(77)
                np.random.seed(77)
                flat = MaskingGenerator((grid, grid), (grid * grid) // 2, min_num_patches=4)().flatten()
                um = np.concatenate([np.zeros((1, 1), np.int32), flat[None, :]], axis=1)
                unk = torch.from_numpy(um).long()
                arrays["unk_mask"] = um.astype(np.int64)
                logits = model(x, unk_mask=unk)
            else:
                logits = model(x, unk_mask=None)
        else:
            logits = model(x)
        loss = torch.nn.CrossEntropyLoss()(logits, y)
        loss.backward()
        for h in hooks:
            h.remove()
        for k, v in model.state_dict().items():
            arrays["p/" + k] = v.detach().numpy().copy()
        for k, v in model.named_parameters():
            arrays["g/" + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy()
        arrays["x"] = x.numpy()
        arrays["y"] = y.numpy()
        arrays["logits"] = logits.detach().numpy()
        arrays["loss"] = loss.detach().numpy()
        arrays["feat"] = feats[0].numpy()
        for i, p in enumerate(probs):
            arrays[f"probs/{i}"] = p.numpy()
        for ck, cv in kw.items():
            if isinstance(cv, str):
                arrays["cfg/" + ck] = np.array(cv)
            else:
                arrays["cfg/" + ck] = np.array(cv, dtype=np.int64)
        arrays["cfg/masked"] = np.array(int(modname == "ViT_mask"), dtype=np.int64)
        arrays["cfg/module"] = np.array(modname)
        path = os.path.join(OUT, f"vit_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(name, "loss", float(loss), "bytes", os.path.getsize(path))

    if not only or "jigsaw" in only:
        gen_jigsaw()
    if not only or "jigsaw_modes" in only:
        gen_jigsaw_modes()
    if not only or "timm" in only:
        gen_timm_vit()
    if not only or "rollout" in only:
        gen_rollout()
    if not only or "grad_rollout" in only:
        gen_grad_rollout()
    if only:
        return
    # ---- mask vectors
    marr = {}
    cases = [(8, 32), (14, 98), (14, 100), (14, 120), (14, 140), (14, 160), (14, 8), (8, 8), (4, 8)]
    idx = 0
    for grid, n in cases:
        for seed in (0, 1, 1234, 1235, 99991):
            random.seed(seed)
            np.random.seed(seed)
            gen = MaskingGenerator((grid, grid), n, min_num_patches=4)
            for rep in range(2):  # two consecutive draws from one RNG stream
                flat = gen().flatten()
                nz = np.nonzero(flat)[0]
                sh = nz.copy()
                np.random.shuffle(sh)
                um = np.concatenate([np.zeros((1, 1), np.int32), flat[None, :]], axis=1)
                t = torch.from_numpy(um).long()
                N = um.shape[1]
                so = (torch.arange(N).unsqueeze(0) * (1 - t) + t * N).numpy()
                marr[f"{idx}/meta"] = np.array([grid, n, seed, rep], dtype=np.int64)
                marr[f"{idx}/flat"] = flat.astype(np.int32)
                marr[f"{idx}/shuffle"] = sh.astype(np.int64)
                marr[f"{idx}/unk_mask"] = um.astype(np.int32)
                marr[f"{idx}/seq_ord"] = so.astype(np.int64)
                idx += 1
    marr["count"] = np.array(idx, dtype=np.int64)
    path = os.path.join(OUT, "masks.npz")
    np.savez_compressed(path, **marr)
    print("masks", idx, "bytes", os.path.getsize(path))


def gen_jigsaw():
    """utils/masking_generator.JigsawPuzzleMaskedRegion('mjp') on seeded images: inputs, RNG seeds and the
    returned (images, masks) for a few geometries -> tests/golden/jigsaw_mjp.npz."""
    from utils.masking_generator import JigsawPuzzleMaskedRegion
    arr = {}
    cases = [(32, 4, 32, 4, 3, 3), (32, 8, 8, 2, 2, 1), (224, 16, 98, 8, 1, 1), (24, 2, 60, 4, 2, 3)]
    for i, (img, p, nmask, nmin, B, C) in enumerate(cases):
        for rep, seed in enumerate((5, 1234)):
            if img == 224 and rep:                 # one draw of the large geometry keeps the fixture small
                continue
            torch.manual_seed(100 + seed)
            x = torch.randn(B, C, img, img)
            random.seed(seed)
            np.random.seed(seed)
            jig = JigsawPuzzleMaskedRegion(img_size=img, patch_size=p, num_masking_patches=nmask, min_num_patches=nmin,
                                           mask_type="mjp")
            out, masks = jig(x.clone())
            k = f"{i}_{rep}"
            arr[k + "/meta"] = np.array([img, p, nmask, nmin, B, C, seed], dtype=np.int64)
            arr[k + "/x"] = x.numpy()
            arr[k + "/out"] = out.numpy()
            arr[k + "/masks"] = masks
    arr["keys"] = np.array(sorted(k[:-5] for k in arr if k.endswith("/meta")))
    path = os.path.join(OUT, "jigsaw_mjp.npz")
    np.savez_compressed(path, **arr)
    print("jigsaw", len(arr["keys"]), "bytes", os.path.getsize(path))


def gen_jigsaw_modes():
    """Every branch of JigsawPuzzleMaskedRegion.__call__ of the top-level masking_generator.py (:144-221) besides
    "mjp" -> tests/golden/jigsaw_modes.npz.  That module imports torchvision and the image-folder reader
    dataloader.public_data at its top, neither usable here, so only the two class definitions (MaskingGenerator,
    JigsawPuzzleMaskedRegion) are compiled out of the file with ast; the name `public_data` the constructor calls is
    bound to a function that returns the synthetic patch bank stored in the fixture (the reader is I/O, the bank is an
    input).  Images, seeds of the three generators the class draws from (random, numpy, torch), outputs and masks."""
    import ast
    import copy
    import math
    path = os.path.join(REF, "masking_generator.py")
    tree = ast.parse(open(path).read(), filename=path)
    names = ["MaskingGenerator", "JigsawPuzzleMaskedRegion"]
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names)
    bank_box = {}
    ns = {"torch": torch, "np": np, "random": random, "math": math, "copy": copy,
          "public_data": lambda **kw: bank_box["bank"].clone()}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    Jig = ns["JigsawPuzzleMaskedRegion"]
    arr, keys, failed = {}, [], {}
    geoms = [(32, 4, 32, 4, 3, 3), (24, 2, 60, 4, 2, 3), (28, 14, 2, 1, 3, 2)]
    for gi, (img, p, nmask, nmin, B, C) in enumerate(geoms):
        L = (img // p) ** 2
        torch.manual_seed(300 + gi)
        bank_box["bank"] = torch.randn(7, C * p * p, L)
        arr[f"g{gi}/bank"] = bank_box["bank"].numpy()
        arr[f"g{gi}/meta"] = np.array([img, p, nmask, nmin, B, C], dtype=np.int64)
        for mode in ("pbf", "orain", "zero", "repeat", "exchange", "Scale", "noise", "pps"):
            for seed in (5, 1234):
                torch.manual_seed(100 + seed)
                x = torch.randn(B, C, img, img) * (40.0 if mode == "noise" else 1.0)    # so that the clip at 0 acts
                jig = Jig(img_size=img, patch_size=p, num_masking_patches=nmask, min_num_patches=nmin, mask_type=mode,
                          channels=C, device=torch.device("cpu"))
                random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
                try:
                    out, masks = jig(x.clone())
                except Exception as e:      # a branch that cannot run in the reference has no vector
                    failed[mode] = f"{type(e).__name__}: {e}"
                    continue
                k = f"g{gi}/{mode}/{seed}"
                keys.append(k)
                arr[k + "/x"] = x.numpy()
                arr[k + "/out"] = out.detach().numpy()
                arr[k + "/masks"] = masks if masks is not None else np.zeros((0,), dtype=np.int32)
    arr["keys"] = np.array(keys)
    out_path = os.path.join(OUT, "jigsaw_modes.npz")
    np.savez_compressed(out_path, **arr)
    print("jigsaw modes", len(keys), "bytes", os.path.getsize(out_path), "branches that raise in the reference:", failed)


def _reference_timm_classes():
    """VisionTransformer (with the reference's unk_mask changes), Block, Attention of tim/models/vision_transformer.py
    and what their constructors call -- PatchEmbed, Mlp, DropPath, to_2tuple, Format, nchw_to, _assert, use_fused_attn --
    from the reference's vendored tim/layers.  The packages themselves cannot be imported (tim/__init__ pulls
    torchvision, and vision_transformer.py imports the pip package timm): the small self-contained files are loaded by
    path, and of the others only the needed class / function definitions are compiled out of their files with ast.
    Every line that runs is the reference's own text."""
    import ast
    import importlib.util
    import logging
    import math
    import typing
    from collections import OrderedDict
    from functools import partial

    import torch.nn.functional as F
    from torch import nn
    from torch.jit import Final
    layers = os.path.join(REF, "tim", "layers")

    def whole(name):
        spec = importlib.util.spec_from_file_location("_ref_tim_" + name, os.path.join(layers, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    helpers, fmt, trace, config = whole("helpers"), whole("format"), whole("trace_utils"), whole("config")
    ns = {"torch": torch, "nn": nn, "F": F, "partial": partial, "logging": logging, "math": math, "Final": Final,
          "OrderedDict": OrderedDict, "to_2tuple": helpers.to_2tuple, "Format": fmt.Format, "nchw_to": fmt.nchw_to,
          "_assert": trace._assert, "use_fused_attn": config.use_fused_attn}
    ns.update({k: getattr(typing, k) for k in ("Callable", "List", "Optional", "Sequence", "Tuple", "Union", "Any")})

    def defs(path, names):
        tree = ast.parse(open(path).read(), filename=path)
        body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
        assert sorted(n.name for n in body) == sorted(names), (path, names)
        exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)

    defs(os.path.join(layers, "drop.py"), ["drop_path", "DropPath"])
    defs(os.path.join(layers, "mlp.py"), ["Mlp"])
    defs(os.path.join(layers, "patch_embed.py"), ["PatchEmbed"])
    defs(os.path.join(REF, "tim", "models", "vision_transformer.py"), ["Attention", "LayerScale", "Block", "VisionTransformer"])
    return ns["VisionTransformer"]


def gen_timm_vit():
    """The reference's timm-style VisionTransformer with unk_mask (trainer.py:172-176 builds vit_small_patch16_224 of
    this class) -> tests/golden/timm_*.npz: state_dict, inputs, logits, loss, every gradient, per-layer attention
    probabilities (hooks on attn_drop with fused_attn off, as trainer.py:81-84 / vit_rollout.py:85-98 do) and the
    token features, for token and avg pooling, masked and unmasked, heads of 16 and of 64."""
    VisionTransformer = _reference_timm_classes()
    torch.set_num_threads(1)
    cases = {
        "timm_token": (dict(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=4), 6, False),
        "timm_token_mask": (dict(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=4), 6, True),
        "timm_avg_mask": (dict(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=4,
                               global_pool="avg"), 6, True),
        "timm_dh64_n197": (dict(img_size=224, patch_size=16, num_classes=7, embed_dim=128, depth=1, num_heads=2), 2, True),
    }
    for name, (kw, B, masked) in cases.items():
        torch.manual_seed(7)
        m = VisionTransformer(weight_init="skip", **kw)
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():                  # non-degenerate values everywhere
            for pn, p in m.named_parameters():
                if pn.endswith("bias") or "cls_token" in pn:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
                elif "norm" in pn and pn.endswith("weight"):
                    p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
                elif "pos_embed" in pn:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * p.shape[-1] ** -0.5)
        for blk in m.blocks:
            blk.attn.fused_attn = False        # trainer.py:81-84 switch_fused_attn
        m.train()
        arrays = {f"param/{k}": v.detach().numpy().copy() for k, v in m.state_dict().items()}
        img = kw["img_size"]
        L = (img // kw["patch_size"]) ** 2
        x = torch.randn(B, 3, img, img, generator=g)
        y = torch.randint(0, kw["num_classes"], (B,), generator=g)
        unk = None
        if masked:
            unk = torch.zeros(1, L + 1, dtype=torch.long)
            unk[0, torch.randperm(L, generator=g)[:L // 2] + 1] = 1
        probs = []
        hooks = [mod.register_forward_hook(lambda mm, i, o: probs.append(o.detach().clone()))
                 for n_, mod in m.named_modules() if "attn_drop" in n_]
        feats = m.forward_features(x, unk)
        logits = m.forward_head(feats)
        for h in hooks:
            h.remove()
        loss = torch.nn.functional.cross_entropy(logits, y)
        loss.backward()
        arrays.update({"x": x.numpy(), "y": y.numpy(), "logits": logits.detach().numpy(), "loss": loss.detach().numpy(),
                       "features": feats.detach().numpy()})
        if unk is not None:
            arrays["unk_mask"] = unk.numpy()
        for i, pr in enumerate(probs):
            arrays[f"probs/{i}"] = pr.numpy()
        for pn, p in m.named_parameters():
            arrays[f"grad/{pn}"] = p.grad.numpy()
        arrays["cfg/heads"] = np.array(kw["num_heads"], dtype=np.int64)
        arrays["cfg/pool"] = np.array(kw.get("global_pool", "token"))
        arrays["cfg/kw"] = np.array(repr(kw))
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(name, "loss", float(loss), "bytes", os.path.getsize(path))


def _reference_functions(path, names):
    """Compile only the named top-level function definitions of a reference file (its module-level imports --
    torchvision, cv2 -- are absent in this image, the functions themselves are pure torch)."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), path
    ns = {"torch": torch}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def gen_rollout():
    """vit_rollout.rollout / last_attn and trainer.rollout on seeded softmax tensors -> tests/golden/rollout.npz."""
    ref_rollout, ref_last = _reference_functions(os.path.join(REF, "vit_rollout.py"), ["rollout", "last_attn"])
    (trainer_rollout,) = _reference_functions(os.path.join(REF, "trainer.py"), ["rollout"])
    arr = {}
    keys = []
    for ci, (B, H, N, L) in enumerate([(2, 3, 17, 3), (2, 3, 65, 3), (1, 3, 197, 2)]):
        torch.manual_seed(500 + ci)
        atts = [torch.softmax(2.0 * torch.randn(B, H, N, N), dim=-1) for _ in range(L)]
        for l, a in enumerate(atts):
            arr[f"c{ci}/att{l}"] = a.numpy()
        arr[f"c{ci}/meta"] = np.array([B, H, N, L], dtype=np.int64)
        for fusion in ("mean", "max", "min"):
            for ratio in (0.9, 0.5, 0.0):
                if N == 197 and (fusion != "mean" or ratio != 0.9):
                    continue
                k = f"c{ci}/{fusion}/{ratio}"
                keys.append(k)
                arr[k + "/roll"] = ref_rollout([a.clone() for a in atts], ratio, fusion, "cpu").numpy()
                arr[k + "/trainer"] = trainer_rollout([a.clone() for a in atts], ratio, fusion).numpy()
                arr[k + "/last"] = ref_last([a.clone() for a in atts], fusion).numpy()
    arr["keys"] = np.array(keys)
    path = os.path.join(OUT, "rollout.npz")
    np.savez_compressed(path, **arr)
    print("rollout", len(keys), "bytes", os.path.getsize(path))


def gen_grad_rollout():
    """vit_grad_rollout.grad_rollout (the function body alone; the file's imports need torchvision / cv2 / PIL) on
    seeded probabilities and gradients of one image -> tests/golden/grad_rollout.npz."""
    path = os.path.join(REF, "vit_grad_rollout.py")
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "grad_rollout"]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    ref = ns["grad_rollout"]
    arr, keys = {}, []
    for ci, (H, N, L) in enumerate([(3, 17, 3), (3, 65, 3), (2, 197, 2)]):
        torch.manual_seed(700 + ci)
        atts = [torch.softmax(2.0 * torch.randn(1, H, N, N), dim=-1) for _ in range(L)]
        grads = [torch.randn(1, H, N, N) * 0.1 for _ in range(L)]
        for l in range(L):
            arr[f"c{ci}/att{l}"] = atts[l].numpy()
            arr[f"c{ci}/grad{l}"] = grads[l].numpy()
        arr[f"c{ci}/meta"] = np.array([H, N, L], dtype=np.int64)
        for ratio in (0.9, 0.5, 0.0):
            k = f"c{ci}/{ratio}"
            keys.append(k)
            arr[k + "/mask"] = ref([a.clone() for a in atts], [g.clone() for g in grads], ratio)
    arr["keys"] = np.array(keys)
    path = os.path.join(OUT, "grad_rollout.npz")
    np.savez_compressed(path, **arr)
    print("grad_rollout", len(keys), "bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
