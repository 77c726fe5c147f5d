"""Attention rollout over the exported attention probabilities, on the device.

Mirrors vit_rollout.py of the reference (rollout :18-61, last_attn :64-73, VITAttentionRollout :77-105)
and the rollout variant trainer.py:17-44 uses in predict().  The reference runs ~10 torch ops and one
batched [N,N]x[N,N] matmul per layer; here a layer is one HIP launch (head fusion, selection of the
int(N*N*discard_ratio) smallest entries, discard, (A+I)/2 with normalised rows) and the product of the L
matrices is one launch that carries only the class-token row through the chain (vitmi_rollout).

Differences a caller can observe: results are always float32; among EQUAL fused values at the discard
threshold the lowest flat indices are dropped first (torch.topk leaves that choice unspecified); an
unknown head_fusion raises ValueError (the reference raises a str, a TypeError in Python 3).
"""
import ctypes

import torch

from . import ops
from ._lib import check, lib

_FUSION = {"mean": 0, "max": 1, "min": 2}


def _fusion_code(head_fusion):
    try:
        return _FUSION[head_fusion]
    except KeyError:
        raise ValueError("Attention head fusion type Not supported") from None


def _rollout(attentions, discard_ratio, head_fusion, keep_first, final_mode):
    fusion = _fusion_code(head_fusion)
    attentions = [a.detach().contiguous() for a in attentions]
    if not attentions:
        raise ValueError("rollout needs at least one attention tensor")
    a0 = attentions[0]
    ops._need_gpu(*attentions)
    if a0.dim() != 4 or a0.shape[-1] != a0.shape[-2]:
        raise ValueError(f"attention tensors must be [B, H, N, N], got {tuple(a0.shape)}")
    for a in attentions:
        if a.shape != a0.shape or a.dtype != a0.dtype or a.device != a0.device:
            raise ValueError("all attention tensors of one rollout must share shape, dtype and device")
    B, H, N, _ = a0.shape
    L = len(attentions)
    k = int(N * N * discard_ratio)                       # vit_rollout.py:34
    if not 0 <= k <= N * N:
        raise ValueError(f"discard_ratio {discard_ratio} outside [0, 1]")
    out = torch.empty((B, N - 1), dtype=torch.float32, device=a0.device)
    ws = ops._ws(lib.vitmi_rollout_workspace_bytes(L, B, N), a0.device)
    ptrs = (ctypes.c_void_p * L)(*[a.data_ptr() for a in attentions])
    with torch.cuda.device(a0.device):
        check(lib.vitmi_rollout(ptrs, ops._dt(a0), L, B, H, N, fusion, k, int(keep_first), final_mode, out.data_ptr(),
                                ws.data_ptr(), ops._stream()), "vitmi_rollout")
    return out


def rollout(attentions, discard_ratio, head_fusion, device=None):
    """vit_rollout.py:18-61: [B, N-1] class-to-patch rollout, divided by its maximum and soft-maxed.  `device` is
    accepted for signature compatibility; the result lives where the attention tensors live."""
    return _rollout(attentions, discard_ratio, head_fusion, keep_first=False, final_mode=1)


def trainer_rollout(attentions, discard_ratio=0.9, head_fusion="mean"):
    """trainer.py:17-44: the variant that never discards flat entry 0 and returns the raw row."""
    return _rollout(attentions, discard_ratio, head_fusion, keep_first=True, final_mode=0)


def last_attn(attentions, head_fusion):
    """vit_rollout.py:64-73: heads-fused class-token row of the last layer, [B, N-1]."""
    fusion = _fusion_code(head_fusion)
    a = attentions[-1].detach().contiguous()
    ops._need_gpu(a)
    B, H, N, _ = a.shape
    out = torch.empty((B, N - 1), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib.vitmi_rollout_last_attn(a.data_ptr(), ops._dt(a), B, H, N, fusion, out.data_ptr(), ops._stream()),
              "vitmi_rollout_last_attn")
    return out


class VITAttentionRollout:
    """vit_rollout.py:77-105: hooks every module whose name contains `attention_layer_name` (the probability taps
    of this package's models are named attn_drop), runs model.forward_features(x, None) and returns the rollout
    ("roll"), the last layer's class row ("attn") or the class-token features ("out")."""

    def __init__(self, model, attention_layer_name="attn_drop", head_fusion="mean", discard_ratio=0.9, device="cuda:0"):
        self.model = model.to(device)
        self.head_fusion = head_fusion
        self.discard_ratio = discard_ratio
        self.device = device
        for name, module in self.model.named_modules():
            if attention_layer_name in name:
                module.register_forward_hook(self.get_attention)
        self.attentions = []
        self.scale = (model.embed_dim // model.blocks[0].attn.num_heads) ** -0.5

    def get_attention(self, module, input, output):
        self.attentions.append(output)

    def __call__(self, input_tensor, ret_out):
        self.attentions = []
        with torch.no_grad():
            output = self.model.forward_features(input_tensor, None)
        if "roll" in ret_out:
            return rollout(self.attentions, self.discard_ratio, self.head_fusion, self.device)
        elif "attn" in ret_out:
            return last_attn(self.attentions, self.head_fusion)
        elif "out" in ret_out:
            return output[:, 0, :]
