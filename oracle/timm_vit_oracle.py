"""CPU oracle for the timm-style VisionTransformer with unk_mask (TEST INFRASTRUCTURE, not
product code).

Pure-torch fp32 restatement, from the source text, of the forward pass in the reference's vendored
tim/models/vision_transformer.py: Attention.forward (:80-102, non-fused branch), Block.forward
(:156-159), _pos_embed (:556-564), forward_features (:615-624), forward_head / forward (:626-637),
with LayerNorm eps 1e-6 (:454) and erf GELU (:455).  Backward is torch autograd over this forward.

PINNED: the module cannot be imported in the build container (tim/__init__ pulls torchvision, and the
file imports the external timm package), so tools/gen_golden.py compiles the class definitions
VisionTransformer, Block, Attention, LayerScale out of the file, and PatchEmbed, Mlp, DropPath and the
few helpers their constructors call out of the reference's vendored tim/layers, and runs them:
tests/golden/timm_*.npz hold state_dict, inputs, logits, loss, every gradient, per-layer probabilities
and token features for token / avg pooling, masked and unmasked, heads of 16 and 64.
tests/test_timm_golden.py holds this restatement to 2e-5 (forward) / 5e-5 (gradients) against them.

Only tests/ may import this module.  Parameters are a flat dict with the vendored class's
state_dict keys: cls_token, pos_embed, patch_embed.proj.{weight,bias},
blocks.<i>.{norm1,norm2}.{weight,bias}, blocks.<i>.attn.{qkv,proj}.{weight,bias},
blocks.<i>.mlp.{fc1,fc2}.{weight,bias}, norm.{weight,bias} (or fc_norm.* with global_pool='avg'),
head.{weight,bias}.
"""
import torch
import torch.nn.functional as F

EPS = 1e-6


def depth_of(params):
    return 1 + max(int(k.split(".")[1]) for k in params if k.startswith("blocks."))


def attention(x, p, pre, heads, collect_probs=None):
    """tim/models/vision_transformer.py:80-102 (the `else` branch: q * scale, softmax, attn @ v)."""
    B, N, C = x.shape
    dh = C // heads
    qkv = F.linear(x, p[pre + "qkv.weight"], p.get(pre + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q = q * dh ** -0.5
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    if collect_probs is not None:
        collect_probs.append(attn)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(out, p[pre + "proj.weight"], p[pre + "proj.bias"])


def mlp(x, p, pre):
    h = F.gelu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"]))
    return F.linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def block(x, p, i, heads, collect_probs=None):
    """tim/models/vision_transformer.py:156-159."""
    pre = f"blocks.{i}."
    C = x.shape[-1]
    x = x + attention(F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], EPS), p, pre + "attn.",
                      heads, collect_probs)
    return x + mlp(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], EPS), p, pre + "mlp.")


def forward_features(p, x, heads, unk_mask=None, collect_probs=None):
    """tim/models/vision_transformer.py:556-564 and :615-624."""
    w = p["patch_embed.proj.weight"]
    tok = F.conv2d(x, w, p["patch_embed.proj.bias"], stride=w.shape[-1]).flatten(2).transpose(1, 2)
    tok = torch.cat((p["cls_token"].expand(tok.shape[0], -1, -1), tok), dim=1)
    pe = p["pos_embed"]
    pos = pe[:, :-1, :]
    if unk_mask is not None:
        unk_embed_index = pe.shape[1] - 1                      # num_patches + 1 (:505)
        seq_ord = torch.arange(tok.size(1)).unsqueeze(0)
        seq_ord = seq_ord * (1 - unk_mask) + unk_mask * unk_embed_index
        pos = pe[:, seq_ord.squeeze(0), :]
    tok = tok + pos
    for i in range(depth_of(p)):
        tok = block(tok, p, i, heads, collect_probs)
    if "norm.weight" in p:
        tok = F.layer_norm(tok, (tok.shape[-1],), p["norm.weight"], p["norm.bias"], EPS)
    return tok


def forward(p, x, heads, unk_mask=None, global_pool="token", collect_probs=None):
    """tim/models/vision_transformer.py:626-637."""
    tok = forward_features(p, x, heads, unk_mask, collect_probs)
    feat = tok[:, 1:].mean(dim=1) if global_pool == "avg" else tok[:, 0]
    if "fc_norm.weight" in p:
        feat = F.layer_norm(feat, (feat.shape[-1],), p["fc_norm.weight"], p["fc_norm.bias"], EPS)
    return F.linear(feat, p["head.weight"], p["head.bias"])


def loss_and_grads(p, x, y, heads, **kw):
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logits = forward(leaf, x, heads, **kw)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return logits.detach(), loss.detach(), grads
