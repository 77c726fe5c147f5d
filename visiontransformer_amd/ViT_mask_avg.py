"""Drop-in for mymodel/ViT_mask_avg.py: VIT.forward(x, unk_mask=None); same names, signatures and state_dict
layout as the reference (masked tokens take the mean of the unmasked patch rows of the positional table)."""
import torch

from ._modules import Attention, FeedForward, PatchEmbed, PreNorm, Transformer, _from_config, pair, parameter
from ._modules import VIT_mask_avg as VIT

__all__ = ["pair", "PatchEmbed", "PreNorm", "FeedForward", "Attention", "Transformer", "VIT", "parameter",
           "creat_VIT", "load_VIT"]


def creat_VIT(config):
    """mymodel/ViT_mask_avg.py creat_VIT: build from config.patch.* (dim_head and pool keep their defaults)."""
    return _from_config(VIT, config)


def load_VIT(model_path, config):
    """mymodel/ViT_mask_avg.py load_VIT: build, then strict-load a state_dict file."""
    model_vit = _from_config(VIT, config)
    model_vit.load_state_dict(torch.load(model_path, weights_only=True))
    return model_vit
