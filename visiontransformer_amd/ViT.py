"""Drop-in for mymodel/ViT.py: same importable names, constructor and forward signatures and
state_dict layout; the arithmetic runs in the gfx950 HIP kernels of libvitmi_hip.so."""
import torch

from ._modules import (Attention, FeedForward, PatchEmbed, PreNorm, Transformer, VIT, _from_config, pair,
                       parameter)

__all__ = ["pair", "PatchEmbed", "PreNorm", "FeedForward", "Attention", "Transformer", "VIT", "parameter",
           "creat_VIT", "load_VIT"]


def creat_VIT(config):
    """mymodel/ViT.py:181-194: build from config.patch.* (dim_head and pool keep their defaults)."""
    return _from_config(VIT, config)


def load_VIT(model_path, config):
    """mymodel/ViT.py:196-211: build, then strict-load a state_dict file."""
    model_vit = _from_config(VIT, config)
    model_vit.load_state_dict(torch.load(model_path, weights_only=True))
    return model_vit
