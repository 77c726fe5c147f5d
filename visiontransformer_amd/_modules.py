"""nn.Module mirror of mymodel/ViT.py and mymodel/ViT_mask.py over the HIP kernels.

Module tree, constructor signatures, attribute names and state_dict keys are the reference's
(mymodel/ViT.py:9-168, mymodel/ViT_mask.py:124-181), so checkpoints load both ways and the
reference's training / attack scripts can construct and call these classes unchanged.  All
arithmetic runs in the gfx950 kernels of libvitmi_hip.so; parameters stay fp32 and
`compute_dtype` (bf16 by default, fp32 for the tight-tolerance mode) selects the activation
type and the kernel instantiation.
"""
import torch
from torch import nn

from . import ops


def pair(t):
    """mymodel/ViT.py:6-7."""
    return t if isinstance(t, tuple) else (t, t)


class _HipModule(nn.Module):
    compute_dtype = torch.bfloat16
    # True: the module computes in fp32 whatever the model's compute dtype (the classifier head: a few rows per image,
    # so exact arithmetic there is free and keeps three bf16 roundings out of the logits and of the head's gradients)
    pin_float32 = False


def set_module_compute_dtype(model, dtype):
    """torch.bfloat16 (MFMA path, default) or torch.float32 (exact-fp32 kernels) on every HIP module of `model`."""
    if dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("compute dtype must be torch.bfloat16 or torch.float32")
    for m in model.modules():
        if isinstance(m, _HipModule):
            m.compute_dtype = torch.float32 if m.pin_float32 else dtype
    return model


class LayerNorm(nn.LayerNorm, _HipModule):
    def forward(self, x):
        if x.dtype != self.compute_dtype:
            x = ops.cast(x, self.compute_dtype) if not x.requires_grad else _CastFn.apply(x, self.compute_dtype)
        return ops.LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class Linear(nn.Linear, _HipModule):
    out_dtype = None  # None = compute dtype

    def forward(self, x):
        return ops.LinearFn.apply(x, self.weight, self.bias, self.compute_dtype, self.out_dtype)


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return ops.cast(g.contiguous(), ctx.src), None


class ProbsTap(nn.Module):
    """Identity module that the attention block calls with the exported [B,H,N,N] probabilities,
    so that ordinary forward hooks observe them.  Stands where the reference has
    Attention.attend = nn.Softmax (mymodel/ViT.py:85,100) and where the timm-style model has
    attn_drop (tim/models/vision_transformer.py:95-96, the module vit_rollout.py:85-87 hooks)."""

    def forward(self, probs):
        # a view when gradients flow, so that backward hooks of either flavour see the gradient of THIS module's
        # input (a hook on an output that is the input itself would be attached to the producer's node)
        return probs.view_as(probs) if probs.requires_grad else probs


def _has_backward_hooks(*mods):
    return any(len(m._backward_hooks) + len(m._backward_pre_hooks) > 0 for m in mods)


def attention_with_probability_gradients(mod, x, norm, w_qkv, b_qkv, proj, heads, taps, cdt):
    """The attention sub-block as the chain LayerNorm -> to_qkv -> softmax(q k^T) -> [taps] -> probs @ v -> to_out of
    separate autograd nodes, so that the gradient reaching the probabilities exists as a tensor and backward hooks
    on the taps receive it (vit_grad_rollout.py:38-56).  Returns (y without residual, tapped probabilities)."""
    if norm is not None:
        x = ops.LayerNormFn.apply(x if x.dtype == cdt else _CastFn.apply(x, cdt), norm.weight, norm.bias, norm.eps)
    qkv = ops.LinearFn.apply(x, w_qkv, b_qkv, cdt, None)
    probs = ops.AttnProbsFn.apply(qkv, heads)
    for t in taps:
        probs = t(probs)
    o = ops.ProbsApplyFn.apply(probs, qkv, heads)
    if proj is not None:
        o = ops.LinearFn.apply(o, proj.weight, proj.bias, cdt, None)
    return o, probs


class PatchEmbed(_HipModule):
    """mymodel/ViT.py:9-46.  Output [B, num_patches, embed_dim] (contiguous, compute dtype)."""

    def __init__(self, image_size=224, patch_size=16, in_c=3, embed_dim=768, norm_layer=None):
        super().__init__()
        image_size = (image_size, image_size)
        patch_size = (patch_size, patch_size)
        self.image_size = image_size
        self.patch_size = patch_size
        self.grid_size = (image_size[0] // patch_size[0], image_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.image_size[0] and W == self.image_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.image_size[0]}*{self.image_size[1]})."
        x = ops.PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, self.compute_dtype)
        return self.norm(x)


class PreNorm(_HipModule):
    """mymodel/ViT.py:52-58.  Runs LayerNorm fused into the wrapped block when it is one of ours."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        if isinstance(self.fn, (Attention, FeedForward)) and not kwargs:
            return self.fn._run(x, self.norm, add_residual=False)
        return self.fn(self.norm(x), **kwargs)


class FeedForward(_HipModule):
    """mymodel/ViT.py:62-73."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(
            Linear(dim, hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            Linear(hidden_dim, dim),
            nn.Dropout(dropout),
        )
        self.drop_p = dropout

    def _run(self, x, norm, add_residual):
        l1, l2 = self.net[0], self.net[3]
        p = self.net[2].p                                   # the two nn.Dropout modules share the constructor's rate
        drop = (p, ops.new_seed(), ops.new_seed()) if p > 0 and self.training else None
        return ops.FFBlockFn.apply(x, norm.weight if norm is not None else None,
                                   norm.bias if norm is not None else None, l1.weight, l1.bias, l2.weight, l2.bias,
                                   norm.eps if norm is not None else 0.0, add_residual, self.compute_dtype, drop)

    def forward(self, x):
        return self._run(x, None, add_residual=False)


class Attention(_HipModule):
    """mymodel/ViT.py:76-104, plus the attributes the timm-style consumers poke
    (num_heads, fused_attn, attn_drop; tim/models/vision_transformer.py:52-102)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.num_heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.fused_attn = True          # False => always export the probabilities
        self.attend = ProbsTap()
        self.attn_drop = ProbsTap()
        self.to_qkv = Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            Linear(inner_dim, dim),
            nn.Dropout(dropout),
        ) if project_out else nn.Identity()
        self.drop_p = dropout
        self.last_probs = None

    def _export_wanted(self):
        return (self.fused_attn is False or len(self.attend._forward_hooks) > 0
                or len(self.attn_drop._forward_hooks) > 0)

    def _run(self, x, norm, add_residual):
        if x.dim() != 3:
            raise ValueError(f"Attention expects [batch, tokens, dim], got {tuple(x.shape)}")
        proj = self.to_out[0] if isinstance(self.to_out, nn.Sequential) else None
        if torch.is_grad_enabled() and _has_backward_hooks(self.attend, self.attn_drop):
            y, self.last_probs = attention_with_probability_gradients(
                self, x, norm, self.to_qkv.weight, None, proj, self.heads, (self.attend, self.attn_drop),
                self.compute_dtype)
            p = self.to_out[1].p if proj is not None and self.training else 0.0
            if p > 0:
                y = ops.DropoutFn.apply(y, p, ops.new_seed())
            if add_residual:
                y = ops.AddFn.apply(y, x if x.dtype == y.dtype else _CastFn.apply(x, y.dtype))
            return y
        want = self._export_wanted()
        # nn.Dropout after to_out (mymodel/ViT.py:90) sits between the projection and the residual add
        p = self.to_out[1].p if proj is not None and self.training else 0.0
        y, probs = ops.AttnBlockFn.apply(
            x, norm.weight if norm is not None else None, norm.bias if norm is not None else None,
            self.to_qkv.weight, proj.weight if proj is not None else None, proj.bias if proj is not None else None,
            self.heads, norm.eps if norm is not None else 0.0, add_residual and p == 0, want, self.compute_dtype)
        if p > 0:
            y = ops.DropoutFn.apply(y, p, ops.new_seed())
            if add_residual:
                y = ops.AddFn.apply(y, x if x.dtype == y.dtype else _CastFn.apply(x, y.dtype))
        if want:
            self.last_probs = self.attn_drop(self.attend(probs))
        return y

    def forward(self, x):
        return self._run(x, None, add_residual=False)


class Transformer(_HipModule):
    """mymodel/ViT.py:107-121: x = attn(x) + x ; x = ff(x) + x, depth times.  Each half-block
    (LayerNorm + projections + core + residual) is one autograd node over fused kernels."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)),
            ]))

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn.fn._run(x, attn.norm, add_residual=True)
            x = ff.fn._run(x, ff.norm, add_residual=True)
        return x


class _BlockView:
    """What `for block in model.blocks: block.attn.fused_attn = False` (trainer.py:81-84,
    mia_attn.py:213-216) and vit_rollout.py:90 expect of one timm-style block."""

    def __init__(self, layer):
        self.attn = layer[0].fn
        self.norm1 = layer[0].norm
        self.mlp = layer[1].fn
        self.norm2 = layer[1].norm


class _VITBase(_HipModule):
    masked = False

    def __init__(self, image_size, patch_size, embed_dim, num_classes, depth, heads, mlp_dim, pool='cls', channels=3,
                 dim_head=8, dropout=0., emb_dropout=0., PE=True):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0
        num_patches = (image_height // patch_height) * (image_width // patch_width)
        assert pool in {'cls', 'mean'}
        self.to_patch_embedding = PatchEmbed(image_size=image_size, patch_size=patch_size, in_c=channels,
                                             embed_dim=embed_dim)
        if self.masked:
            self.unk_embed_index = num_patches + 1
        self.PE = True          # the reference ignores its PE argument (ViT.py:136); callers set it afterwards
        self._make_pos_embedding(num_patches, embed_dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, embed_dim))
        self.dropout = nn.Dropout(emb_dropout)
        if self.masked:
            self.train_method = 'zero'
        self.transformer = Transformer(embed_dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(
            LayerNorm(embed_dim),
            Linear(embed_dim, num_classes),
        )
        self.mlp_head[1].out_dtype = torch.float32
        for m in self.mlp_head:                 # [B, D] rows: LayerNorm and the classifier run in fp32
            m.pin_float32 = True
            m.compute_dtype = torch.float32
        self.embed_dim = embed_dim
        self.num_tokens = num_patches + 1

    def _make_pos_embedding(self, num_patches, embed_dim):
        if self.masked:
            self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 2, embed_dim))
        else:
            self.pos_embedding = nn.Parameter(torch.randn(num_patches + 1, embed_dim))

    # ---- runtime switches ---------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """torch.bfloat16 (MFMA path, default) or torch.float32 (exact-fp32 kernels)."""
        return set_module_compute_dtype(self, dtype)

    @staticmethod
    def set_gelu(mode):
        """"fast" (default) or "erf": see visiontransformer_amd.ops.set_gelu (a process-wide switch of the library)."""
        return ops.set_gelu(mode)

    # ---- timm-style aliases used by trainer.py / mia_attn.py / vit_rollout.py ------------------
    @property
    def blocks(self):
        return [_BlockView(layer) for layer in self.transformer.layers]

    # ---- forward ------------------------------------------------------------------------
    def _seq_ord(self, n, unk_mask, device):
        return None

    def _pe_table(self, n, unk_mask, device):
        """(table [rows, D], seq_ord or None): token i receives table[seq_ord[i]] (row i without seq_ord)."""
        return self.pos_embedding.view(-1, self.pos_embedding.shape[-1]), self._seq_ord(n, unk_mask, device)

    def forward_features(self, x, unk_mask=None):
        """Token sequence after the last block, [B, N, D] (timm naming; vit_rollout.py:98)."""
        tok = self.to_patch_embedding(x)
        n = tok.shape[1] + 1
        pe2d, seq_ord = self._pe_table(n, unk_mask, tok.device) if bool(self.PE) else (None, None)
        x = ops.TokensFn.apply(tok, self.cls_token, pe2d, seq_ord)
        x = self.dropout(x)
        return self.transformer(x)

    def _head(self, x):
        x = ops.MeanPoolFn.apply(x) if self.pool == 'mean' else x[:, 0]
        x = self.to_latent(x)
        return self.mlp_head(x)


class VIT(_VITBase):
    """mymodel/ViT.py:124-168."""
    masked = False

    def forward(self, x):
        return self._head(self.forward_features(x))


class VIT_mask(_VITBase):
    """mymodel/ViT_mask.py:124-181 (exported as ViT_mask.VIT)."""
    masked = True

    def _seq_ord(self, n, unk_mask, device):
        # ViT_mask.py:159-170.  The table has N+1 rows, the last one is the learned "unknown"
        # slot; without a mask the first N rows are used in order.
        if unk_mask is None:
            return None
        if self.train_method == 'mask':
            return ops.seq_ord(unk_mask, n, self.unk_embed_index, device)
        if self.train_method == 'zero':
            # the reference's 'zero' branch reads an undefined attribute (ViT_mask.py:169); its
            # evident intent -- masked tokens receive a zero positional row -- is implemented
            return ops.seq_ord(unk_mask, n, -1, device)                  # negative index = no positional row
        return None

    def forward(self, x, unk_mask=None):
        return self._head(self.forward_features(x, unk_mask))


def _masked_order(n, unk_mask, unk_index, device):
    m = unk_mask.to(device=device, dtype=torch.long).reshape(-1)
    return m, ops.seq_ord(m, n, unk_index, device)


class VIT_mask_avg(_VITBase):
    """mymodel/ViT_mask_avg.py:123-182 (exported as ViT_mask_avg.VIT): a 2-D table [N, D]; with an
    unk_mask every masked token receives the MEAN of the unmasked patch rows (the class row excluded,
    :167-168).  The [N+1, D] table with that extra row is assembled by ordinary tensor ops on the
    device (it is N x D numbers); the gather-add over the batch and its backward are the HIP kernels."""
    masked = False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.unk_embed_index = self.num_tokens          # num_patches + 1

    def _pe_table(self, n, unk_mask, device):
        pe = self.pos_embedding
        if unk_mask is None:
            return pe, None
        m, order = _masked_order(n, unk_mask, self.unk_embed_index, device)
        keep = (m[1:] == 0).to(pe.dtype)                               # unmasked patch rows
        unk_pos = (pe[1:] * keep[:, None]).sum(0, keepdim=True) / keep.sum()
        return torch.cat([pe, unk_pos], 0), order

    def forward(self, x, unk_mask=None):
        return self._head(self.forward_features(x, unk_mask))


class VIT_ape(_VITBase):
    """mymodel/ViT_ape.py:144-198 (exported as ViT_ape.VIT): the positional table is the fixed
    sinusoid of :11-32, not a parameter (absent from the state_dict, as in the reference); masked
    tokens receive no positional row.  The reference's masked branch reads an undefined attribute
    (:186); its evident intent, a zero row, is what runs here."""
    masked = False

    def _make_pos_embedding(self, num_patches, embed_dim):
        import math
        n = num_patches + 1
        pe = torch.zeros(n, embed_dim)
        position = torch.arange(0, n, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, embed_dim, 2).float() * (-math.log(10000.0) / embed_dim))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pos_embedding", pe.unsqueeze(0), persistent=False)
        self.unk_embed_index = num_patches + 1

    def _pe_table(self, n, unk_mask, device):
        pe2d = self.pos_embedding.view(-1, self.pos_embedding.shape[-1])
        if unk_mask is None:
            return pe2d, None
        return pe2d, ops.seq_ord(unk_mask, n, -1, device)              # negative index = no positional row

    def forward(self, x, unk_mask=None):
        return self._head(self.forward_features(x, unk_mask))


class parameter(nn.Module):
    """mymodel/ViT.py:172-178: learnable additive image perturbation (unused by the callers)."""

    def __init__(self):
        super().__init__()
        self.para = nn.Parameter(torch.randn(1, 3, 32, 32))

    def forward(self, x):
        return x + self.para


def _from_config(cls, config):
    return cls(
        image_size=config.patch.image_size,
        patch_size=config.patch.patch_size,
        num_classes=config.patch.num_classes,
        embed_dim=config.patch.embed_dim,
        depth=config.patch.depth,
        heads=config.patch.heads,
        mlp_dim=config.patch.mlp_dim,
        dropout=config.patch.dropout,
        emb_dropout=config.patch.emb_dropout,
        channels=config.patch.channels,
    )
