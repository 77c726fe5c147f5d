"""Attention rollout over the exported attention probabilities, on the device.

Mirrors vit_rollout.py of the reference (rollout :18-61, last_attn :64-73, VITAttentionRollout :77-105),
the rollout variant trainer.py:17-44 uses in predict(), and vit_grad_rollout.py (grad_rollout :9-36,
VITAttentionGradRollout :38-66).  The reference runs ~10 torch ops and one
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
    """The collector of vit_rollout.py:77-105, same constructor and call signature.  A forward hook on every module
    whose qualified name contains `attention_layer_name` (this package's probability taps are called attn_drop)
    gathers the [B,H,N,N] probabilities of one `model.forward_features(x, None)` pass, run without autograd; what
    comes back depends on a substring of `ret_out`: "roll" the rollout, "attn" the last layer's class row, "out" the
    class-token features."""

    def __init__(self, model, attention_layer_name="attn_drop", head_fusion="mean", discard_ratio=0.9, device="cuda:0"):
        self.model = model.to(device)
        self.device = device
        self.head_fusion, self.discard_ratio = head_fusion, discard_ratio
        self.attentions = []
        taps = [m for name, m in self.model.named_modules() if attention_layer_name in name]
        for tap in taps:
            tap.register_forward_hook(self.get_attention)
        heads = model.blocks[0].attn.num_heads
        self.scale = (model.embed_dim // heads) ** -0.5       # kept: callers of the reference class read it

    def get_attention(self, module, input, output):
        self.attentions.append(output)

    def __call__(self, input_tensor, ret_out):
        self.attentions = []
        with torch.no_grad():
            tokens = self.model.forward_features(input_tensor, None)
        if "roll" in ret_out:
            return rollout(self.attentions, self.discard_ratio, self.head_fusion, self.device)
        if "attn" in ret_out:
            return last_attn(self.attentions, self.head_fusion)
        if "out" in ret_out:
            return tokens[:, 0, :]
        return None


def grad_rollout(attentions, gradients, discard_ratio):
    """vit_grad_rollout.py:9-36: rollout of probabilities weighted by the gradients that reached them.  Per layer
    the heads are fused as the mean of probability x gradient with negative means zeroed, the
    int(N*N*discard_ratio) smallest entries dropped, and (A + I)/2 normalised the way the reference's
    `a / a.sum(dim=-1)` does on its batch of one (entry (i, j) by the sum of row j); the class row of the product,
    divided by its maximum, comes back as a numpy [w, w] map for one image ([B, w, w] for a batch, every image
    treated as the reference treats its single one)."""
    if len(attentions) != len(gradients) or not attentions:
        raise ValueError("grad_rollout needs one gradient per attention tensor")
    a0 = attentions[0]
    attentions = [a.detach().to(a0.device).contiguous() for a in attentions]
    gradients = [g.detach().to(device=a0.device, dtype=a0.dtype).contiguous() for g in gradients]
    ops._need_gpu(*attentions)
    if a0.dim() != 4 or a0.shape[-1] != a0.shape[-2]:
        raise ValueError(f"attention tensors must be [B, H, N, N], got {tuple(a0.shape)}")
    for a, g in zip(attentions, gradients):
        if a.shape != a0.shape or g.shape != a0.shape or a.dtype != a0.dtype:
            raise ValueError("all tensors of one gradient rollout must share shape and dtype")
    B, H, N, _ = a0.shape
    L = len(attentions)
    k = int(N * N * discard_ratio)
    if not 0 <= k <= N * N:
        raise ValueError(f"discard_ratio {discard_ratio} outside [0, 1]")
    width = int((N - 1) ** 0.5)
    if width * width != N - 1:
        raise ValueError(f"{N - 1} patches do not form a square map")
    out = torch.empty((B, N - 1), dtype=torch.float32, device=a0.device)
    ws = ops._ws(lib.vitmi_rollout_workspace_bytes(L, B, N), a0.device)
    ptrs = (ctypes.c_void_p * L)(*[a.data_ptr() for a in attentions])
    gptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in gradients])
    with torch.cuda.device(a0.device):
        check(lib.vitmi_grad_rollout(ptrs, gptrs, ops._dt(a0), L, B, H, N, k, out.data_ptr(), ws.data_ptr(),
                                     ops._stream()), "vitmi_grad_rollout")
    mask = out.reshape(B, width, width).cpu().numpy()
    return mask[0] if B == 1 else mask


class VITAttentionGradRollout:
    """vit_grad_rollout.py:38-66.  Remembers what every probability tap of the model (modules whose name contains
    `attention_layer_name`) lets through on the way forward and which gradient comes back through it, for a loss
    that is the logit of one category; `__call__` returns grad_rollout of the two lists.  Registering backward hooks
    on the taps is what makes the attention blocks run with the probabilities as a tensor of the autograd graph.
    The tensors stay on the device (the reference moves each to the CPU as it arrives).  Autograd delivers the
    gradients last layer first and the reference zips the two lists as they come, pairing layer l's probabilities
    with layer L-1-l's gradient; that pairing is kept (match_layers=False, the default); match_layers=True pairs
    every layer with its own gradient."""

    def __init__(self, model, attention_layer_name='attn_drop', discard_ratio=0.9, match_layers=False):
        self.model = model
        self.discard_ratio = discard_ratio
        self.match_layers = match_layers
        self._taps = [m for n, m in model.named_modules() if attention_layer_name in n]
        for tap in self._taps:
            tap.register_forward_hook(self.get_attention)
            tap.register_full_backward_hook(self.get_attention_gradient)
        self.attentions = []
        self.attention_gradients = []

    def get_attention(self, module, input, output):
        self.attentions.append(output.detach())

    def get_attention_gradient(self, module, grad_input, grad_output):
        self.attention_gradients.append(grad_input[0].detach())

    def __call__(self, input_tensor, category_index):
        self.attentions, self.attention_gradients = [], []
        self.model.zero_grad()
        output = self.model(input_tensor)
        category_mask = torch.zeros_like(output)
        category_mask[:, category_index] = 1
        (output * category_mask).sum().backward()
        grads = self.attention_gradients[::-1] if self.match_layers else self.attention_gradients
        return grad_rollout(self.attentions, grads, self.discard_ratio)
