"""Host-side block-mask sampler that produces the `unk_mask` input of ViT_mask.VIT.forward.

Mirrors the interface of masking_generator.MaskingGenerator of the reference
(masking_generator.py:13-84; same class at utils/masking_generator.py:29-100) and consumes
the global `random` state in the same order, so a seeded run yields the same indices bit for
bit.  The sampling itself is tiny integer work and stays on the host, as in the reference;
the device-side index arithmetic it feeds lives in the token-assembly kernel.
"""
import math
import random

import numpy as np


class MaskingGenerator:
    def __init__(self, input_size, num_masking_patches, min_num_patches=8, max_num_patches=None, min_aspect=0.3,
                 max_aspect=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.init_num_masking_patches = num_masking_patches
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def __repr__(self):
        return "Generator(%d, %d -> [%d ~ %d], max = %d, %.3f ~ %.3f)" % (
            self.height, self.width, self.min_num_patches, self.max_num_patches, self.num_masking_patches,
            self.log_aspect_ratio[0], self.log_aspect_ratio[1])

    def get_shape(self):
        return self.height, self.width

    def _mask(self, mask, max_mask_patches):
        """Try up to ten rectangles; paint the first one that adds between 1 and
        max_mask_patches new cells.  Returns the number of cells added."""
        for _ in range(10):
            target_area = random.uniform(self.min_num_patches, max_mask_patches)
            aspect_ratio = math.exp(random.uniform(*self.log_aspect_ratio))
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if not (w < self.width and h < self.height):
                continue
            top = random.randint(0, self.height - h)
            left = random.randint(0, self.width - w)
            window = mask[top:top + h, left:left + w]
            fresh = h * w - int(window.sum())
            if 0 < fresh <= max_mask_patches:
                window[...] = 1
                return fresh
        return 0

    def __call__(self):
        mask = np.zeros(shape=self.get_shape(), dtype=np.int32)
        mask_count = 0
        while mask_count < self.num_masking_patches:
            room = min(self.num_masking_patches - mask_count, self.max_num_patches)
            delta = self._mask(mask, room)
            if delta == 0:
                break
            mask_count += delta
        return mask

    def _num_patches_decaying(self, total_epoch=300, cur_epoch=0, masking_ratio=-1):
        cur_ratio = masking_ratio if masking_ratio > 0 else 0.1
        self.num_masking_patches = int(self.num_patches * cur_ratio)

    # ---- helpers around the sampler (masking_generator.py:131-136, :169-170) ------------------
    def masked_indexes(self):
        """(flat mask, indices of masked patches, a shuffled copy drawn from np.random)."""
        mask = self().flatten()
        nonzero = np.nonzero(mask)[0]
        shuffled = nonzero.copy()
        np.random.shuffle(shuffled)
        return mask, nonzero, shuffled

    @staticmethod
    def unk_mask_from_flat(flat_mask):
        """[1, L+1] int32 with a leading 0 for the class token, as JigsawPuzzleMaskedRegion returns."""
        return np.concatenate([np.zeros((1, 1), dtype=np.int32), flat_mask[np.newaxis, :].astype(np.int32)], axis=1)


def seq_ord_from_unk_mask(unk_mask, unk_embed_index=None):
    """Integer restatement of ViT_mask.py:162-163 on the host (int64 [1, N])."""
    m = np.asarray(unk_mask).astype(np.int64).reshape(1, -1)
    n = m.shape[1]
    u = n if unk_embed_index is None else int(unk_embed_index)
    return np.arange(n, dtype=np.int64)[None, :] * (1 - m) + m * u
