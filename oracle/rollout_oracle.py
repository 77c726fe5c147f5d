"""CPU oracle for attention rollout (TEST INFRASTRUCTURE, not product code).

Restates, op for op in torch on the CPU, the two rollout functions of the reference:
  vit_rollout.py:18-61  rollout(attentions, discard_ratio, head_fusion, device)  and :64-73 last_attn
  trainer.py:17-44      rollout(attentions, discard_ratio=0.9, head_fusion="mean")
including their quirks: the discard runs over the FLATTENED [N*N] fused matrix of each sample
(k = int(N*N*discard_ratio) smallest entries, not per row); trainer.py takes the candidates from flat[:, 1:]
and shifts the indices by one, so flat entry 0 (class-to-class) is never discarded; the matrices are
multiplied in full and row 0 is taken at the end.

Pinning: tests/golden/rollout.npz holds outputs of the reference's own function bodies.  vit_rollout.py and
trainer.py cannot be imported here (their top-level imports need torchvision / cv2), so tools/gen_golden.py
parses the two files with `ast`, compiles ONLY the `rollout` / `last_attn` function definitions (pure torch)
and runs those on seeded inputs.

grad_rollout restates vit_grad_rollout.py:9-36 (gradient-weighted rollout; pinned the same way by
tests/golden/grad_rollout.npz).  Its quirks, kept: the function is written for ONE image -- `flat[0, indices] = 0`
discards in sample 0 only and `a / a.sum(dim=-1)` (no keepdim) divides entry (i, j) by the sum of row j; the result
is a [w, w] numpy array divided by its maximum.
"""
import numpy as np
import torch


def fuse(attention, head_fusion):
    if head_fusion == "mean":
        return attention.mean(axis=1)
    if head_fusion == "max":
        return attention.max(axis=1)[0]
    if head_fusion == "min":
        return attention.min(axis=1)[0]
    raise ValueError("Attention head fusion type Not supported")


def normalised_layer(attention, discard_ratio, head_fusion, keep_first):
    """One loop body: fused, discarded, (A+I)/2 row-normalised [B,N,N] (vit_rollout.py:22-44 / trainer.py:21-40)."""
    fused = fuse(attention.float(), head_fusion).clone()
    flat = fused.view(fused.size(0), -1)
    k = int(flat.size(-1) * discard_ratio)
    if keep_first:
        _, indices = flat[:, 1:].topk(k, -1, False)
        indices = indices + 1
    else:
        _, indices = flat.topk(k, -1, False)
    flat.scatter_(1, indices.to(torch.int64), 0)
    eye = torch.eye(fused.size(-1))
    a = (fused + 1.0 * eye) / 2
    return a / a.sum(dim=-1)[..., None]


def rollout(attentions, discard_ratio, head_fusion):
    """vit_rollout.py:18-61 -> [B, N-1]."""
    result = torch.eye(attentions[0].size(-1))
    for attention in attentions:
        result = torch.matmul(normalised_layer(attention, discard_ratio, head_fusion, False), result)
    mask = result[:, 0, 1:]
    mask = mask / mask.max(-1)[0][..., None]
    return torch.nn.functional.softmax(mask, dim=-1)


def trainer_rollout(attentions, discard_ratio=0.9, head_fusion="mean"):
    """trainer.py:17-44 -> [B, N-1]."""
    result = torch.eye(attentions[0].size(-1))
    for attention in attentions:
        result = torch.matmul(normalised_layer(attention, discard_ratio, head_fusion, True), result)
    return result[:, 0, 1:]


def last_attn(attentions, head_fusion):
    """vit_rollout.py:64-73."""
    return fuse(attentions[-1].float(), head_fusion)[:, 0, 1:]


def grad_rollout(attentions, gradients, discard_ratio):
    """vit_grad_rollout.py:9-36 for a batch of one -> numpy [w, w]."""
    n = attentions[0].size(-1)
    result = torch.eye(n)
    for attention, grad in zip(attentions, gradients):
        if attention.size(0) != 1:
            raise ValueError("grad_rollout is defined for one image")
        fused = (attention.float() * grad.float()).mean(axis=1).clone()
        fused[fused < 0] = 0
        flat = fused.view(1, -1)
        _, indices = flat.topk(int(flat.size(-1) * discard_ratio), -1, False)
        flat[0, indices] = 0
        a = (fused + 1.0 * torch.eye(n)) / 2
        a = a / a.sum(dim=-1)                 # [1,N,N] / [1,N]: column j divided by the sum of row j
        result = torch.matmul(a, result)
    mask = result[0, 0, 1:]
    width = int(mask.size(-1) ** 0.5)
    mask = mask.reshape(width, width).numpy()
    return mask / np.max(mask)
