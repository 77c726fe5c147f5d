"""CPU oracle for the mask-index path (TEST INFRASTRUCTURE, not product code).

Restates, with explicit loops, the BEiT block-mask sampler that feeds
ViT_mask.VIT.forward's `unk_mask` argument:
  masking_generator.py:13-84 (identical text at utils/masking_generator.py:29-100)
  masking_generator.py:131-136 (_get_masked_indexes) and :169-170 (cls padding).

Random draws come from the GLOBAL `random` and `numpy.random` states, consumed in
the reference's order: per attempt uniform(area), uniform(log-aspect), then --
only when the rectangle fits -- randint(top), randint(left); after the mask is
complete one np.random.shuffle over a copy of the non-zero indices.

Pinned by tests/golden/masks.npz, captured from the reference's importable
utils/masking_generator.MaskingGenerator.
"""
import math
import random

import numpy as np


def block_mask(height, width, num_masking_patches, min_num_patches=4, max_num_patches=None,
               min_aspect=0.3, max_aspect=None):
    """One call of MaskingGenerator.__call__ (masking_generator.py:64-77 inner loop,
    :80-93 outer loop).  Returns int32 [height, width]; the count of ones may be
    below num_masking_patches when ten attempts in a row add nothing."""
    max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
    max_aspect = max_aspect or 1 / min_aspect
    lo, hi = math.log(min_aspect), math.log(max_aspect)
    mask = np.zeros((height, width), dtype=np.int32)
    count = 0
    while count < num_masking_patches:
        budget = min(num_masking_patches - count, max_num_patches)
        added = 0
        for _ in range(10):
            area = random.uniform(min_num_patches, budget)
            aspect = math.exp(random.uniform(lo, hi))
            h = int(round(math.sqrt(area * aspect)))
            w = int(round(math.sqrt(area / aspect)))
            if w < width and h < height:
                top = random.randint(0, height - h)
                left = random.randint(0, width - w)
                already = 0
                for i in range(top, top + h):
                    for j in range(left, left + w):
                        already += int(mask[i, j])
                if 0 < h * w - already <= budget:
                    for i in range(top, top + h):
                        for j in range(left, left + w):
                            if mask[i, j] == 0:
                                mask[i, j] = 1
                                added += 1
            if added > 0:
                break
        if added == 0:
            break
        count += added
    return mask


def masked_indexes(height, width, num_masking_patches, min_num_patches=4):
    """masking_generator.py:131-136: (flat mask, nonzero, shuffled copy of nonzero)."""
    flat = block_mask(height, width, num_masking_patches, min_num_patches).flatten()
    nz = np.nonzero(flat)[0]
    shuffled = nz.copy()
    np.random.shuffle(shuffled)
    return flat, nz, shuffled


def unk_mask_from_flat(flat):
    """masking_generator.py:169-170: prepend the always-visible cls position -> [1, L+1] int32."""
    return np.concatenate([np.zeros((1, 1), dtype=np.int32), flat[np.newaxis, :].astype(np.int32)], axis=1)


def seq_ord(unk_mask, unk_embed_index=None):
    """mymodel/ViT_mask.py:162-163 on integers: i*(1-m[i]) + m[i]*unk_index, unk_index = N."""
    m = np.asarray(unk_mask).astype(np.int64)
    n = m.shape[-1]
    u = n if unk_embed_index is None else unk_embed_index
    return np.arange(n, dtype=np.int64)[None, :] * (1 - m) + m * u
