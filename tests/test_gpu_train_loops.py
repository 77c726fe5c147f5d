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
