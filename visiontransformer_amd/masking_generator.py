"""Host-side block-mask sampler that produces the `unk_mask` input of ViT_mask.VIT.forward, and the
jigsaw image augmentation built on it (index sampling on the host, patch movement in one HIP gather).

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


class JigsawPuzzleMaskedRegion:
    """The augmentation trainer.py:126 applies before model(inputs, unk_mask=...): same constructor
    arguments, RNG consumption and return values as masking_generator.py:99-199 (its "mjp" branch is also
    utils/masking_generator.py:156-171), with the image side as one HIP gather kernel.

    Where the reference unfolds the batch, overwrites the masked columns and folds it back (three full
    passes plus a deep copy for "mjp"), every branch here is a per-patch source table `src` [L] uploaded
    to the device and one vitmi_jigsaw_gather launch that reads and writes the batch once:
      "mjp"    masked patch nonzero[k] <- patch nonzero_shuffle[k] of the same image
      "pbf"    masked patch nonzero[k] <- patch nonzero_shuffle[k] of public image pub_idx[b]
      "orain"  images returned untouched (only the mask is sampled)
      "zero"   masked patches zeroed; returns (images, None)
      "repeat" masked patches <- mean over seq_ord of the image's patches (masked positions count patch 0,
               as the reference's arange*(1-mask) index does); returns (images, None)
      "exchange" the "mjp" shuffle, returned as (images, None)
    and the whole-image branches, which sample no mask (masking_generator.py:210-221):
      "Scale"  images * s, s = torch.rand(1) * 1.0 + 0.5 from torch's CPU generator; (images, None)
      "noise"  clip(images + N(0, 0.1), 0, 255), the normal draw taken from numpy's global generator on the host
               exactly as the reference takes it, added and clipped by one launch; (images, None)
      anything else: (images, None) untouched, like the reference's final else.

    pub_data is the unfolded public bank [P, C*p*p, L] that dataloader.public_data returns (dataloader.py:17-37).
    Decoding an image folder needs torchvision and is not part of this package: pass the bank as a tensor
    (`pub_data=`), or, for "pbf" without it, the constructor calls the CALLER's own `dataloader.public_data(
    pub_data_dir, ...)` (the module trainer.py:196-203 runs beside) and raises if there is none.  There is no CPU
    path: images must be device tensors.
    """

    MODES = ("mjp", "pbf", "orain", "zero", "repeat", "exchange", "Scale", "noise")

    def __init__(self, img_size=224, patch_size=16, num_masking_patches=8, min_num_patches=2, mask_type="mjp",
                 pub_data_dir="../data/ImageNet100/public/", channels=3, device=None, pub_data=None):
        import torch
        self.img_size, self.patch_size, self.channels = int(img_size), int(patch_size), int(channels)
        grid = int(img_size // patch_size)
        self.grid = grid
        self.masking_generator = MaskingGenerator((grid, grid), num_masking_patches=num_masking_patches,
                                                  min_num_patches=min_num_patches)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else torch.device("cpu")
        self.scale = [0.5, 1.5]
        self.mask_type = mask_type
        self.pub_data_dir = pub_data_dir
        self.pub_data = None
        self._bank = None
        if pub_data is not None:
            self.set_public_data(pub_data)
        elif mask_type == "pbf":
            raise ValueError("mask_type 'pbf' needs the public patch bank: pass pub_data=[P, C*p*p, L] (the tensor the "
                             f"reference builds with dataloader.public_data(root_dir={pub_data_dir!r}, ...), "
                             "dataloader.py:17-37); this package reads no image folders")

    def set_public_data(self, pub_data):
        """Keep the bank as the reference holds it (self.pub_data, [P, C*p*p, L]) and as [P, C, H, W] images, the
        layout the gather reads in whole patch rows."""
        p, g, c = self.patch_size, self.grid, self.channels
        if pub_data.dim() != 3 or pub_data.shape[1] != c * p * p or pub_data.shape[2] != g * g:
            raise ValueError(f"pub_data must be [P, {c * p * p}, {g * g}], got {tuple(pub_data.shape)}")
        self.pub_data = pub_data.to(self.device)
        P = pub_data.shape[0]
        self._bank = (self.pub_data.view(P, c, p, p, g, g).permute(0, 1, 4, 2, 5, 3)
                      .reshape(P, c, g * p, g * p).contiguous())

    def _update_masking_generator(self, total_epoch, epoch, masking_ratio=-1):
        self.masking_generator._num_patches_decaying(total_epoch, epoch, masking_ratio)

    def _get_masked_indexes(self):
        return self.masking_generator.masked_indexes()

    def _get_batch_masked_indexes(self):
        return MaskingGenerator.unk_mask_from_flat(self.masking_generator().flatten())

    def __call__(self, x):
        """x: [N, C, H, W] float32 or bfloat16 on the device.  Returns (images, masks) like the reference: masks is
        the [1, L+1] int32 numpy array with the class-token pad, or None for "zero" / "repeat"."""
        from . import ops
        if self.mask_type == "Scale":                                       # masking_generator.py:210-213
            import torch
            scale = torch.rand(1) * (self.scale[1] - self.scale[0]) + self.scale[0]
            return ops.scale_add_clip(x, scale=float(scale)), None
        if self.mask_type == "noise":                                       # masking_generator.py:215-218
            noise = np.random.normal(0, 0.1, tuple(x.shape)).astype(np.float32)
            return ops.scale_add_clip(x, addend=ops.upload(noise, x.device), lo=0.0, hi=255.0), None
        if self.mask_type not in ("mjp", "pbf", "orain", "zero", "repeat", "exchange"):
            return x, None                                                  # the reference's final else
        flat, nz, sh = self._get_masked_indexes()
        masks = MaskingGenerator.unk_mask_from_flat(flat)
        if self.mask_type == "orain":
            return x, masks
        ops._need_gpu(x)
        N, C, H, W = x.shape
        p, L = self.patch_size, self.grid * self.grid
        if H != self.img_size or W != self.img_size:
            raise ValueError(f"images are {H}x{W}, the generator was built for {self.img_size}")
        table = np.empty(L + N, dtype=np.int32)                              # src [L] then bank_idx [N]: one upload
        src = table[:L]
        src[:] = np.arange(L, dtype=np.int32)
        table[L:] = 0
        bank = fill = None
        if self.mask_type in ("mjp", "exchange"):
            src[nz] = sh
            if self.mask_type == "exchange":
                masks = None
        elif self.mask_type == "pbf":
            if self._bank is None:
                raise ValueError("mask_type 'pbf' needs set_public_data() first")
            P = self._bank.shape[0]
            pub_idx = np.random.choice(np.arange(P), N, replace=False)      # raises when N > P, as the reference does
            src[nz] = L + sh
            bank = self._bank if self._bank.dtype == x.dtype else self._bank.to(x.dtype)
            table[L:] = pub_idx
        elif self.mask_type == "zero":
            src[nz] = -1
            masks = None
        else:                                                               # "repeat"
            weight = (1 - flat).astype(np.float32)
            weight[0] = float(len(nz)) + float(1 - flat[0])
            fill = ops.patch_weighted_mean(x, p, ops.upload(weight, x.device))
            src[nz] = -2
            masks = None
        dev_table = ops.upload(table, x.device)
        out = ops.jigsaw_gather(x, p, dev_table[:L], bank, dev_table[L:] if bank is not None else None, fill)
        return out, masks
