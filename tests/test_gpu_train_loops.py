"""The reference's training-step idioms around the drop-in modules, as its scripts write them:
defence_method/PEdrop.py:108-125 (jigsaw augmentation, fp16 autocast context, GradScaler, torch.optim.Adam, an
eval pass under set_grad_enabled(False)) and trainer.py:117-142 (plain fp32 step with the mask drawn per batch).
These check that the modules behave as ordinary nn.Modules inside those loops; numbers are checked for being
finite and for learning (the per-tensor parity tests live in test_gpu_model.py / test_gpu_timm_vit.py)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_pedrop_style_step_on_the_timm_style_model():
    from torch.cuda.amp import GradScaler, autocast
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    from visiontransformer_amd.vision_transformer import VisionTransformer
    torch.manual_seed(0), random.seed(0), np.random.seed(0)
    model = VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=4).to(DEV)
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)
    jigsaw = JigsawPuzzleMaskedRegion(img_size=32, patch_size=4, num_masking_patches=24, min_num_patches=4,
                                      mask_type="mjp", device=DEV)
    scaler = GradScaler()
    inputs = torch.randn(32, 3, 32, 32, device=DEV)
    labels = torch.randint(0, 10, (32,), device=DEV)
    losses = []
    model.train()
    for _ in range(12):
        x, unk_mask = jigsaw(inputs)
        unk_mask = torch.from_numpy(unk_mask).long().to(DEV)
        optimizer.zero_grad()
        with torch.set_grad_enabled(True):
            with autocast():
                outputs = model(x, unk_mask=unk_mask)
                loss = criterion(outputs, labels)
                scaler.scale(loss).backward()
                scaler.step(optimizer)
                scaler.update()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.2, losses
    model.eval()
    with torch.set_grad_enabled(False):
        with autocast():
            outputs = model(inputs, unk_mask=None)
            _, preds = torch.max(outputs, 1)
    assert outputs.shape == (32, 10) and not outputs.requires_grad and preds.shape == (32,)


def test_trainer_style_step_with_data_parallel_wrappers_absent():
    """trainer.py:117-142 with ViT_mask.VIT: mask per batch, train_method as trainer.py:180 sets it, Adam, and the
    same model evaluated with unk_mask=None afterwards."""
    from visiontransformer_amd import ViT_mask
    from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
    torch.manual_seed(1), random.seed(1), np.random.seed(1)
    model = ViT_mask.VIT(32, 4, 96, 10, 2, 6, 128).to(DEV)
    model.train_method = "mask"
    jigsaw = JigsawPuzzleMaskedRegion(img_size=32, patch_size=4, num_masking_patches=20, min_num_patches=4,
                                      mask_type="orain", device=DEV)
    optimizer = torch.optim.Adam(model.parameters(), lr=2e-3, betas=(0.9, 0.99))
    criterion = torch.nn.CrossEntropyLoss()
    inputs = torch.randn(48, 3, 32, 32, device=DEV)
    labels = torch.randint(0, 10, (48,), device=DEV)
    losses = []
    model.train()
    for _ in range(12):
        x, unk_mask = jigsaw(inputs)
        unk_mask = torch.from_numpy(unk_mask).long().to(DEV)
        optimizer.zero_grad()
        outputs = model(x, unk_mask=unk_mask)
        loss = criterion(outputs, labels)
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.2, losses
    model.eval()
    with torch.no_grad():
        assert model(inputs, unk_mask=None).shape == (48, 10)


def test_graphed_forward_backward_matches_eager():
    """One HIP graph of forward + loss + backward (visiontransformer_amd.graphs) leaves the same gradients and, with
    the optimizer outside the graph, the same parameters as the eager step, bit for bit, with a new mask per step."""
    from visiontransformer_amd import ViT_mask
    from visiontransformer_amd.graphs import GraphedForwardBackward
    from visiontransformer_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    kw = dict(image_size=32, patch_size=4, embed_dim=64, num_classes=10, depth=2, heads=4, mlp_dim=128)
    gen = torch.Generator().manual_seed(0)
    xs = [torch.randn(16, 3, 32, 32, generator=gen).to(dev) for _ in range(4)]
    ys = [torch.randint(0, 10, (16,), generator=gen).to(dev) for _ in range(4)]
    masks = []
    for _ in range(4):
        um = torch.zeros(1, 65, dtype=torch.long)
        um[0, torch.randperm(64, generator=gen)[:32] + 1] = 1
        masks.append(um.to(dev))
    ce = torch.nn.functional.cross_entropy

    def make():
        torch.manual_seed(3)
        m = ViT_mask.VIT(**kw).to(dev).train()
        m.train_method = "mask"
        return m, FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.99))

    m1, o1 = make()
    m2, o2 = make()
    g = GraphedForwardBackward(m2, ce, xs[0], ys[0], masks[0])
    for x, y, um in zip(xs, ys, masks):
        o1.zero_grad()
        l1 = ce(m1(x, unk_mask=um), y)
        l1.backward()
        l2 = g(x, y, um)
        assert torch.equal(l1.detach(), l2.detach())
        for (k, a), b in zip(m1.named_parameters(), m2.parameters()):
            assert torch.equal(a.grad, b.grad), k
        o1.step()
        o2.step()
    assert all(torch.equal(a, b) for a, b in zip(m1.parameters(), m2.parameters()))
    # an optimizer that changes the parameters through torch ops (their version moves, the cached compute copies
    # go stale) and a cleared cache in between: the graph must still compute with the current weights
    from visiontransformer_amd import ops

    def make_sgd():
        torch.manual_seed(4)
        m = ViT_mask.VIT(**kw).to(dev).train()
        m.train_method = "mask"
        return m, torch.optim.SGD(m.parameters(), lr=0.05)

    m4, o4 = make_sgd()
    m5, o5 = make_sgd()
    g5 = GraphedForwardBackward(m5, ce, xs[0], ys[0], masks[0])
    for i, (x, y, um) in enumerate(zip(xs, ys, masks)):
        o4.zero_grad()
        l4 = ce(m4(x, unk_mask=um), y)
        l4.backward()
        l5 = g5(x, y, um)
        assert torch.equal(l4.detach(), l5.detach()), i
        o4.step()
        o5.step()
        if i == 1:
            ops.clear_cast_cache()
    assert all(torch.equal(a, b) for a, b in zip(m4.parameters(), m5.parameters()))
    m3 = ViT_mask.VIT(dropout=0.1, **kw).to(dev).train()
    with pytest.raises(ValueError, match="dropout"):
        GraphedForwardBackward(m3, ce, xs[0], ys[0], masks[0])
