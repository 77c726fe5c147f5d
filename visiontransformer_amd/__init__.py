"""visiontransformer_amd: the ViT encoder hot path of mrkize/VisionTransformer
(mymodel/ViT.py, mymodel/ViT_mask.py) as hand-written gfx950 (MI355X) HIP kernels behind the
reference's own nn.Module interface.

    from visiontransformer_amd import ViT, ViT_mask
    model = ViT.VIT(224, 16, 768, 100, 12, 12, 3072, dim_head=64).cuda()
    logits = model(images)

Importing the package loads libvitmi_hip.so; without it the import fails (there is no
fallback compute path).
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)
from . import ViT, ViT_ape, ViT_mask, ViT_mask_avg, dpsgd, ops  # noqa: F401
from .masking_generator import MaskingGenerator  # noqa: F401
from .ops import set_gelu  # noqa: F401

__version__ = "0.1.0"
