"""Drop-in for mymodel/ViT_mask.py: VIT.forward(x, unk_mask=None) with the positional table's
extra "unknown" row; same names, signatures and state_dict layout as the reference."""
import torch

from ._modules import Attention, FeedForward, PatchEmbed, PreNorm, Transformer, _from_config, pair, parameter
from ._modules import VIT_mask as VIT

__all__ = ["pair", "PatchEmbed", "PreNorm", "FeedForward", "Attention", "Transformer", "VIT", "parameter",
           "creat_VIT", "load_VIT"]


def creat_VIT(config):
    """mymodel/ViT_mask.py:194-207."""
    return _from_config(VIT, config)


def load_VIT(model_path, config):
    """mymodel/ViT_mask.py:209-223."""
    model_vit = _from_config(VIT, config)
    model_vit.load_state_dict(torch.load(model_path, weights_only=True))
    return model_vit
