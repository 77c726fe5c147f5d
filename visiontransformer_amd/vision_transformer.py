"""timm-style `VisionTransformer` with the `unk_mask` positional-embedding switch, as the reference
vendors it (tim/models/vision_transformer.py:52-159, 382-637) and as trainer.py, PEdrop.py and
mia_attn.py instantiate it (`vit_small_patch16_224`, tim/models/vision_transformer.py:1498-1504).

Same module tree, attribute names and state_dict keys as the vendored class (cls_token,
pos_embed [1, tokens + 1, D] whose last row is the learned "unknown" slot, patch_embed.proj,
blocks.<i>.{norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2}, norm, head); the arithmetic runs
in the gfx950 kernels of libvitmi_hip.so: qkv projection with bias, scale folded into the score
product (tim: q * scale before q k^T, line 93 -- the same number up to one rounding), LayerNorm
eps 1e-6, a final norm before the head.

Only the configuration the reference's scripts use is implemented; the other timm switches
(qk_norm, layer scale, pre_norm, stochastic depth, patch dropout, no_embed_class, models without
a class token) raise NotImplementedError instead of silently computing something else.
"""
from functools import partial

import torch
from torch import nn

from . import ops
from ._modules import (LayerNorm, Linear, PatchEmbed as _PatchEmbedBase, ProbsTap, _CastFn, _HipModule, set_module_compute_dtype,
                       _has_backward_hooks, attention_with_probability_gradients)

__all__ = ["Attention", "Mlp", "Block", "PatchEmbed", "VisionTransformer", "vit_small_patch16_224",
           "vit_base_patch16_224", "vit_tiny_patch16_224"]


class PatchEmbed(_PatchEmbedBase):
    """tim/layers PatchEmbed as VisionTransformer uses it: Conv2d(k = s = patch) + flatten, bias on."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        if not bias:
            raise NotImplementedError("patch embedding without bias (pre_norm models) is not implemented")
        super().__init__(image_size=img_size, patch_size=patch_size, in_c=in_chans, embed_dim=embed_dim)
        self.img_size = self.image_size


class Attention(_HipModule):
    """tim/models/vision_transformer.py:52-102."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0., proj_drop=0.,
                 norm_layer=nn.LayerNorm, fuse_attn=False):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        if qk_norm:
            raise NotImplementedError("qk_norm is not implemented in the HIP path")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = True          # False => export the probabilities on every forward
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.attn_drop = ProbsTap()     # hooked by vit_rollout.py:85-87; called with [B,H,N,N] probabilities
        self.attn_drop.p = attn_drop
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.drop_p = max(attn_drop, proj_drop)
        self.last_probs = None

    def _export_wanted(self):
        return self.fused_attn is False or len(self.attn_drop._forward_hooks) > 0

    def _run(self, x, norm, add_residual):
        p = self.proj_drop.p if self.training else 0.0       # proj_drop sits between the projection and the residual
        drop_probs = self.attn_drop.p > 0 and self.training
        if drop_probs or (torch.is_grad_enabled() and _has_backward_hooks(self.attn_drop)):
            # dropout on the probabilities (tim/models/vision_transformer.py:95-96) and backward hooks on attn_drop
            # (vit_grad_rollout.py:44-46) both need the probabilities as a tensor of the autograd graph
            taps = (self.attn_drop,)
            if drop_probs:
                pd, seed = self.attn_drop.p, ops.new_seed()
                taps = (lambda t: ops.DropoutFn.apply(t, pd, seed), self.attn_drop)
            y, self.last_probs = attention_with_probability_gradients(
                self, x, norm, self.qkv.weight, self.qkv.bias, self.proj, self.num_heads, taps, self.compute_dtype)
            if p > 0:
                y = ops.DropoutFn.apply(y, p, ops.new_seed())
            if add_residual:
                y = ops.AddFn.apply(y, x if x.dtype == y.dtype else _CastFn.apply(x, y.dtype))
            return y
        want = self._export_wanted()
        y, probs = ops.AttnBlockFn.apply(
            x, norm.weight if norm is not None else None, norm.bias if norm is not None else None,
            self.qkv.weight, self.proj.weight, self.proj.bias, self.num_heads,
            norm.eps if norm is not None else 0.0, add_residual and p == 0, want, self.compute_dtype, self.qkv.bias)
        if p > 0:
            y = ops.DropoutFn.apply(y, p, ops.new_seed())
            if add_residual:
                y = ops.AddFn.apply(y, x if x.dtype == y.dtype else _CastFn.apply(x, y.dtype))
        if want:
            self.last_probs = self.attn_drop(probs)
        return y

    def forward(self, x):
        return self._run(x, None, add_residual=False)


class Mlp(_HipModule):
    """timm Mlp: fc1 -> GELU(erf) -> drop -> fc2 -> drop."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("only the erf GELU is implemented in the HIP path")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)
        self.drop_p = drop

    def _run(self, x, norm, add_residual):
        p = self.drop1.p
        drop = (p, ops.new_seed(), ops.new_seed()) if p > 0 and self.training else None
        return ops.FFBlockFn.apply(x, norm.weight if norm is not None else None,
                                   norm.bias if norm is not None else None, self.fc1.weight, self.fc1.bias,
                                   self.fc2.weight, self.fc2.bias, norm.eps if norm is not None else 0.0,
                                   add_residual, self.compute_dtype, drop)

    def forward(self, x):
        return self._run(x, None, add_residual=False)


class Block(_HipModule):
    """tim/models/vision_transformer.py:115-159: x + attn(norm1(x)); x + mlp(norm2(x)).  Each half is
    one autograd node over fused kernels (LayerNorm, projections, core, residual)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_norm=False, proj_drop=0., attn_drop=0.,
                 init_values=None, drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, mlp_layer=Mlp):
        super().__init__()
        if init_values:
            raise NotImplementedError("layer scale (init_values) is not implemented in the HIP path")
        if drop_path > 0.:
            raise NotImplementedError("stochastic depth (drop_path) is not implemented in the HIP path")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_norm=qk_norm, attn_drop=attn_drop,
                              proj_drop=proj_drop, norm_layer=norm_layer)
        self.ls1 = nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer,
                             drop=proj_drop)
        self.ls2 = nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = self.attn._run(x, self.norm1, add_residual=True)
        return self.mlp._run(x, self.norm2, add_residual=True)


class VisionTransformer(_HipModule):
    """tim/models/vision_transformer.py:382-637 (forward(x, unk_mask=None))."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, global_pool='token',
                 embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True, qk_norm=False, init_values=None,
                 class_token=True, no_embed_class=False, pre_norm=False, fc_norm=None, drop_rate=0., pos_drop_rate=0.,
                 patch_drop_rate=0., proj_drop_rate=0., attn_drop_rate=0., drop_path_rate=0., weight_init='',
                 embed_layer=PatchEmbed, norm_layer=None, act_layer=None, block_fn=Block, mlp_layer=Mlp):
        super().__init__()
        assert global_pool in ('', 'avg', 'token')
        assert class_token or global_pool != 'token'
        for flag, name in ((not class_token, "class_token=False"), (no_embed_class, "no_embed_class"),
                           (pre_norm, "pre_norm"), (patch_drop_rate > 0, "patch_drop_rate > 0"),
                           (drop_path_rate > 0, "drop_path_rate > 0"), (qk_norm, "qk_norm"),
                           (bool(init_values), "init_values")):
            if flag:
                raise NotImplementedError(f"VisionTransformer({name}) is not implemented in the HIP path")
        if norm_layer is not None or (act_layer is not None and act_layer is not nn.GELU):
            raise NotImplementedError("custom norm_layer / act_layer are not implemented in the HIP path")
        use_fc_norm = global_pool == 'avg' if fc_norm is None else fc_norm
        norm_layer = partial(LayerNorm, eps=1e-6)
        act_layer = nn.GELU

        self.num_classes = num_classes
        self.global_pool = global_pool
        self.num_features = self.embed_dim = embed_dim
        self.num_prefix_tokens = 1
        self.no_embed_class = False
        self.grad_checkpointing = False

        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                       embed_dim=embed_dim, bias=True)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        embed_len = num_patches + self.num_prefix_tokens
        self.pos_embed = nn.Parameter(torch.randn(1, embed_len + 1, embed_dim) * .02)
        self.pos_drop = nn.Dropout(p=pos_drop_rate)
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        self.fuse_attn = False
        self.blocks = nn.Sequential(*[
            block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_norm=qk_norm,
                     init_values=init_values, proj_drop=proj_drop_rate, attn_drop=attn_drop_rate, drop_path=0.,
                     norm_layer=norm_layer, act_layer=act_layer, mlp_layer=mlp_layer)
            for _ in range(depth)])
        self.norm = norm_layer(embed_dim) if not use_fc_norm else nn.Identity()
        self.unk_embed_index = num_patches + 1
        self.fc_norm = norm_layer(embed_dim) if use_fc_norm else nn.Identity()
        self.head_drop = nn.Dropout(drop_rate)
        self.head = Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self._pin_head()
        if weight_init != 'skip':
            self.init_weights(weight_init)

    def init_weights(self, mode=''):
        """init_weights_vit_timm (tim/models/vision_transformer.py:518-525, 640-648): truncated normal
        (std 0.02) Linear weights, zero biases, pos_embed std 0.02, cls_token std 1e-6."""
        assert mode in ('jax', 'jax_nlhb', 'moco', '')
        if mode:
            raise NotImplementedError("only the default timm weight init is implemented")
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _pin_head(self):
        # the modules that only ever see the pooled [B, D] rows compute in fp32 and the logits leave in fp32
        for m in (self.fc_norm, self.head):
            if isinstance(m, _HipModule):
                m.pin_float32 = True
                m.compute_dtype = torch.float32
        if isinstance(self.head, Linear):
            self.head.out_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        """torch.bfloat16 (MFMA path, default) or torch.float32 (exact-fp32 kernels)."""
        return set_module_compute_dtype(self, dtype)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=None):
        self.num_classes = num_classes
        if global_pool is not None:
            assert global_pool in ('', 'avg', 'token')
            self.global_pool = global_pool
        self.head = Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self._pin_head()

    def _seq_ord(self, n, unk_mask, device):
        # tim/models/vision_transformer.py:556-564: masked positions read the last table row
        if unk_mask is None:
            return None
        return ops.seq_ord(unk_mask, n, self.unk_embed_index, device)

    def _pos_embed(self, tok, unk_mask=None):
        """cls prepend + positional rows (the first N of the table, or gathered through seq_ord)."""
        n = tok.shape[1] + 1
        x = ops.TokensFn.apply(tok, self.cls_token, self.pos_embed.view(-1, self.pos_embed.shape[-1]),
                               self._seq_ord(n, unk_mask, tok.device))
        return self.pos_drop(x)

    def forward_features(self, x, unk_mask=None):
        x = self.patch_embed(x)
        x = self._pos_embed(x, unk_mask)
        x = self.blocks(x)
        return self.norm(x)

    def forward_head(self, x, pre_logits=False):
        if self.global_pool:
            if self.global_pool == 'avg':
                x = ops.MeanPoolFn.apply(x[:, self.num_prefix_tokens:].contiguous())
            else:
                x = x[:, 0]
        x = self.fc_norm(x)
        x = self.head_drop(x)
        return x if pre_logits else self.head(x)

    def forward(self, x, unk_mask=None):
        if self.global_pool == 'token' and isinstance(self.norm, LayerNorm):
            # LayerNorm works row by row and only the class row is read: normalise that row alone, in fp32
            # (tim/models/vision_transformer.py:618-633 normalises all rows and then drops all but one)
            x = self.blocks(self._pos_embed(self.patch_embed(x), unk_mask))[:, 0]
            if x.dtype != torch.float32:
                x = _CastFn.apply(x, torch.float32) if x.requires_grad else ops.cast(x, torch.float32)
            x = ops.LayerNormFn.apply(x.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps)
            x = self.head_drop(self.fc_norm(x))
            return self.head(x)
        return self.forward_head(self.forward_features(x, unk_mask))


def _create(pretrained, model_args, kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not available here: load a state_dict instead")
    return VisionTransformer(**dict(model_args, **kwargs))


def vit_tiny_patch16_224(pretrained=False, **kwargs):
    return _create(pretrained, dict(patch_size=16, embed_dim=192, depth=12, num_heads=3), kwargs)


def vit_small_patch16_224(pretrained=False, **kwargs):
    """tim/models/vision_transformer.py:1498-1504 (the model trainer.py:174-176 builds)."""
    return _create(pretrained, dict(patch_size=16, embed_dim=384, depth=12, num_heads=6), kwargs)


def vit_base_patch16_224(pretrained=False, **kwargs):
    return _create(pretrained, dict(patch_size=16, embed_dim=768, depth=12, num_heads=12), kwargs)
