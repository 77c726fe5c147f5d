"""CPU oracle for the image half of the jigsaw augmentation (TEST INFRASTRUCTURE, not product code).

Restates JigsawPuzzleMaskedRegion.__call__ of the reference in numpy:
  masking_generator.py:144-221 (every branch); the "mjp" branch is the same text as
  utils/masking_generator.py:156-171.
Every branch is: Unfold(p, stride p) -> overwrite the masked columns -> Fold back, followed by the
class-token padding of the mask.  Unfold/Fold for kernel = stride are restated as the index map
  patches[n, (c*p + ky)*p + kx, gy*gw + gx] = x[n, c, gy*p + ky, gx*p + kx]        (torch.nn.Unfold).

Random draws follow the reference's order on the GLOBAL `random` / `numpy.random` states: the block
mask (masking_oracle.masked_indexes: python `random`, then one np.random.shuffle) and, for "pbf"
only, one np.random.choice(arange(P), N, replace=False) afterwards (masking_generator.py:162-165).

Pinning: the "mjp" branch is pinned by tests/golden/jigsaw_mjp.npz, captured from the reference's
importable utils/masking_generator.JigsawPuzzleMaskedRegion.  The module that holds the other
branches (masking_generator.py at the reference's root) imports torchvision, which this image does
not have; tools/gen_golden.py compiles its two class definitions alone (MaskingGenerator,
JigsawPuzzleMaskedRegion) out of the file and hands the constructor a synthetic patch bank in place of
the image-folder reader, and tests/golden/jigsaw_modes.npz holds what every other branch ("pbf",
"orain", "zero", "repeat", "exchange", "Scale", "noise", an unknown name) returned: all PINNED, bit
for bit except "repeat" (the fp32 mean is summed in another order: 2e-6).
"""
import numpy as np

from . import masking_oracle


def unfold(x, p):
    """torch.nn.Unfold((p, p), stride=p) on [N, C, H, W] -> [N, C*p*p, L]."""
    n, c, h, w = x.shape
    gh, gw = h // p, w // p
    out = np.empty((n, c * p * p, gh * gw), dtype=x.dtype)
    for ch in range(c):
        for ky in range(p):
            for kx in range(p):
                out[:, (ch * p + ky) * p + kx, :] = x[:, ch, ky::p, kx::p].reshape(n, gh * gw)
    return out


def fold(patches, c, h, w, p):
    """torch.nn.Fold(output_size=(h, w), kernel_size=(p, p), stride=p) on [N, C*p*p, L] (no overlaps, so a
    plain scatter)."""
    n = patches.shape[0]
    gh, gw = h // p, w // p
    out = np.empty((n, c, h, w), dtype=patches.dtype)
    for ch in range(c):
        for ky in range(p):
            for kx in range(p):
                out[:, ch, ky::p, kx::p] = patches[:, (ch * p + ky) * p + kx, :].reshape(n, gh, gw)
    return out


def apply(x, p, mask_type, flat, nonzero, shuffled, pub_data=None, pub_idx=None):
    """The image transform for indices already drawn.  x: float array [N, C, H, W]; pub_data: the unfolded
    public bank [P, C*p*p, L] (dataloader.py:17-37).  Returns the new image batch (x itself for "orain")."""
    n, c, h, w = x.shape
    if mask_type == "orain":                                   # masking_generator.py:175-179
        return x
    cols = unfold(x, p)
    if mask_type == "mjp":                                     # :148-158
        before = cols.copy()
        cols[:, :, nonzero] = before[:, :, shuffled]
    elif mask_type == "pbf":                                   # :159-172
        cols[:, :, nonzero] = pub_data[pub_idx, :, :][:, :, shuffled]
    elif mask_type == "zero":                                  # :182-188
        cols[:, :, nonzero] = 0
    elif mask_type == "repeat":                                # :191-199
        seq_ord = np.arange(cols.shape[2]) * (1 - flat)
        avg = cols[:, :, seq_ord].astype(np.float32).mean(axis=2, dtype=np.float32)[:, :, None]
        cols[:, :, nonzero] = avg.astype(cols.dtype)
    else:
        raise ValueError(mask_type)
    return fold(cols, c, h, w, p)


def jigsaw(x, p, num_masking_patches, min_num_patches, mask_type, pub_data=None):
    """One JigsawPuzzleMaskedRegion.__call__: draws from the global RNG states, returns (images, masks);
    masks is the [1, L+1] int32 array with the class-token pad, or None for the branches that return none
    (masking_generator.py:181-221).  "Scale" draws its factor from torch's CPU generator, "noise" its normal field
    from numpy's; neither samples a mask, nor does an unknown mask_type (returned untouched)."""
    if mask_type == "Scale":                                                # :210-213
        import torch
        scale = torch.rand(1) * (1.5 - 0.5) + 0.5
        return (torch.from_numpy(x) * scale).numpy(), None
    if mask_type == "noise":                                                # :215-218
        noisy = x + np.random.normal(0, 0.1, np.shape(x)).astype(np.float32)
        return np.clip(noisy, 0, 255), None
    if mask_type not in ("mjp", "pbf", "orain", "zero", "repeat", "exchange"):
        return x, None
    n, c, h, w = x.shape
    flat, nz, sh = masking_oracle.masked_indexes(h // p, w // p, num_masking_patches, min_num_patches)
    pub_idx = None
    if mask_type == "pbf":
        pub_idx = np.random.choice(np.arange(pub_data.shape[0]), n, replace=False).tolist()
    out = apply(x, p, "mjp" if mask_type == "exchange" else mask_type, flat, nz, sh, pub_data, pub_idx)
    masks = None if mask_type in ("zero", "repeat", "exchange") else masking_oracle.unk_mask_from_flat(flat)
    return out, masks
