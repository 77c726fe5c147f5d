"""CPU oracle for the ViT encoder hot path (TEST INFRASTRUCTURE, not product code).

A functional, pure-torch fp32/fp64 restatement of the forward pass that the
reference defines in mymodel/ViT.py and mymodel/ViT_mask.py, and of the positional-embedding
variants mymodel/ViT_mask_avg.py (masked tokens take the mean of the unmasked patch rows; 2-D
table + unk_mask) and mymodel/ViT_ape.py (fixed sinusoidal table; no pos_embedding parameter).  Backward is
obtained by torch autograd over this forward, exactly as the reference does.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

Parity pinning: tests/test_oracle_golden.py checks every function here against
fixtures under tests/golden/ that tools/gen_golden.py captured by importing the
reference modules themselves (reference @ /root/reference, torch 2.10 CPU fp32).

Parameters are passed as a flat dict using the reference's state_dict keys
(mymodel/ViT.py:124-149): pos_embedding, cls_token,
to_patch_embedding.proj.{weight,bias}, transformer.layers.<i>.0.norm.*,
transformer.layers.<i>.0.fn.to_qkv.weight, transformer.layers.<i>.0.fn.to_out.0.*,
transformer.layers.<i>.1.norm.*, transformer.layers.<i>.1.fn.net.{0,3}.*,
mlp_head.{0,1}.*
"""
import math

import torch


def patch_embed(x, weight, bias):
    """mymodel/ViT.py:36-46: Conv2d(kernel=stride=p) then flatten(2).transpose(1,2).

    Restated as a GEMM over non-overlapping patches: token (gy,gx) of image b is
    the dot product of weight[d, c, i, j] with x[b, c, gy*p+i, gx*p+j].
    """
    B, C, H, W = x.shape
    D, Cw, p, p2 = weight.shape
    assert C == Cw and p == p2
    gh, gw = H // p, W // p
    patches = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
    return patches @ weight.reshape(D, C * p * p).t() + bias


def layer_norm(x, gamma, beta, eps=1e-5):
    """mymodel/ViT.py:55 (nn.LayerNorm default eps=1e-5, biased variance, affine)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * gamma + beta


def gelu_erf(x):
    """mymodel/ViT.py:67: nn.GELU() default = exact erf form."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def attention(x, w_qkv, w_out, b_out, heads):
    """mymodel/ViT.py:93-104.  Returns (output, post-softmax probabilities [B,H,N,N]).

    to_qkv has no bias (:86); the scale dim_head**-0.5 multiplies the scores
    AFTER the q.k matmul (:98, :83); to_out is absent (Identity) when
    heads == 1 and dim_head == dim (:80, :88-91), signalled here by w_out=None.
    """
    B, N, _ = x.shape
    inner = w_qkv.shape[0] // 3
    dh = inner // heads
    qkv = x @ w_qkv.t()
    q, k, v = (t.reshape(B, N, heads, dh).permute(0, 2, 1, 3) for t in qkv.split(inner, dim=-1))
    dots = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    probs = torch.softmax(dots, dim=-1)
    out = (probs @ v).permute(0, 2, 1, 3).reshape(B, N, inner)
    if w_out is not None:
        out = out @ w_out.t() + b_out
    return out, probs


def feed_forward(x, w1, b1, w2, b2, keep_hidden=None, keep_out=None):
    """mymodel/ViT.py:62-73: Linear -> GELU(erf) -> Dropout -> Linear -> Dropout.  A dropout is stated as the
    multiplier tensor (0 or 1/(1-p) per element) the caller supplies; None = identity (p = 0 or eval mode)."""
    h = gelu_erf(x @ w1.t() + b1)
    if keep_hidden is not None:
        h = h * keep_hidden
    y = h @ w2.t() + b2
    return y * keep_out if keep_out is not None else y


def sinusoid_table(n, d, max_len=5000):
    """mymodel/ViT_ape.py:11-32 (PositionalEncoding.pe[0, :n]): sin on even, cos on odd columns."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe[:n]


def seq_ord_from_mask(unk_mask, unk_embed_index):
    """mymodel/ViT_mask.py:162-163: masked positions re-index to the unknown slot."""
    n = unk_mask.shape[-1]
    ar = torch.arange(n, dtype=unk_mask.dtype).unsqueeze(0)
    return ar * (1 - unk_mask) + unk_mask * unk_embed_index


def depth_of(params):
    d = 0
    while f"transformer.layers.{d}.0.norm.weight" in params:
        d += 1
    return d


def transformer_block(x, params, i, heads, collect_probs=None, keep=None):
    """mymodel/ViT.py:117-121: x = attn(LN(x)) + x ; x = ff(LN(x)) + x.  keep: optional dropout multipliers
    {(i, "attn_out"): [B,N,D] (ViT.py:90), (i, "hidden"): [B,N,M] (:68), (i, "ff_out"): [B,N,D] (:70)}."""
    keep = keep or {}
    pre = f"transformer.layers.{i}."
    w_out = params.get(pre + "0.fn.to_out.0.weight")
    b_out = params.get(pre + "0.fn.to_out.0.bias")
    a, probs = attention(layer_norm(x, params[pre + "0.norm.weight"], params[pre + "0.norm.bias"]),
                         params[pre + "0.fn.to_qkv.weight"], w_out, b_out, heads)
    if collect_probs is not None:
        collect_probs.append(probs)
    if (i, "attn_out") in keep:
        a = a * keep[(i, "attn_out")]
    x = a + x
    f = feed_forward(layer_norm(x, params[pre + "1.norm.weight"], params[pre + "1.norm.bias"]),
                     params[pre + "1.fn.net.0.weight"], params[pre + "1.fn.net.0.bias"],
                     params[pre + "1.fn.net.3.weight"], params[pre + "1.fn.net.3.bias"],
                     keep.get((i, "hidden")), keep.get((i, "ff_out")))
    return f + x


def forward_features(params, x, heads, unk_mask=None, train_method="mask", use_pe=True,
                     collect_probs=None, keep=None):
    """Token sequence after the last block, [B,N,D].

    mymodel/ViT.py:151-162 when pos_embedding is 2-D [N,D];
    mymodel/ViT_mask.py:152-175 when it is 3-D [1,N+1,D] (last row = unknown slot).
    """
    tok = patch_embed(x, params["to_patch_embedding.proj.weight"], params["to_patch_embedding.proj.bias"])
    B = tok.shape[0]
    tok = torch.cat([params["cls_token"].expand(B, -1, -1), tok], dim=1)
    if "pos_embedding" not in params:                   # ViT_ape.py:162,178-187: fixed sinusoidal table
        n, d = tok.shape[1], tok.shape[2]
        pos = sinusoid_table(n, d).unsqueeze(0)
        if unk_mask is not None:
            # the reference reads an undefined attribute here (ViT_ape.py:186); the evident
            # intent is a zero row for masked positions
            order = seq_ord_from_mask(unk_mask, n).squeeze(0)
            pos = torch.cat([pos, torch.zeros(1, 1, d)], dim=1)[:, order, :]
        pe = None
    else:
        pe = params["pos_embedding"]
    if pe is None:
        pass
    elif pe.dim() == 2 and unk_mask is not None:        # ViT_mask_avg.py:165-171
        n = tok.shape[1]
        unmask_idx = torch.nonzero((1 - unk_mask).squeeze(0)).squeeze(1).unsqueeze(0)
        unk_pos = pe[unmask_idx[:, 1:]].mean(1)         # mean over the unmasked PATCH rows (cls excluded)
        order = seq_ord_from_mask(unk_mask, n)
        pos = torch.cat([pe, unk_pos], 0)[order]
    elif pe.dim() == 2:                                 # ViT.py:158-159, ViT_mask_avg.py:173
        pos = pe
    else:                                               # ViT_mask.py:159-170
        n = tok.shape[1]
        pos = pe[:, :-1, :]
        if unk_mask is not None:
            order = seq_ord_from_mask(unk_mask, n).squeeze(0)
            if train_method == "mask":
                pos = pe[:, order, :]
            elif train_method == "zero":
                # the reference line (:169) reads an undefined attribute; the evident
                # intent is a zero row in the unknown slot
                tbl = torch.cat([pos, torch.zeros(1, 1, pe.shape[-1], dtype=pe.dtype)], dim=1)
                pos = tbl[:, order, :]
    if use_pe:
        tok = tok + pos
    for i in range(depth_of(params)):
        tok = transformer_block(tok, params, i, heads, collect_probs, keep)
    return tok


def forward(params, x, heads, pool="cls", unk_mask=None, train_method="mask", use_pe=True,
            collect_probs=None, keep=None):
    """Logits [B,num_classes]: mymodel/ViT.py:164-168 on top of forward_features."""
    tok = forward_features(params, x, heads, unk_mask, train_method, use_pe, collect_probs, keep)
    feat = tok.mean(dim=1) if pool == "mean" else tok[:, 0]
    feat = layer_norm(feat, params["mlp_head.0.weight"], params["mlp_head.0.bias"])
    return feat @ params["mlp_head.1.weight"].t() + params["mlp_head.1.bias"]


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss(), mean reduction (trainer.py:188)."""
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.gather(1, labels[:, None]).squeeze(1)).mean()


def loss_and_grads(params, x, y, heads, **kw):
    """One training-step's worth of numbers: (logits, loss, {name: grad})."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits = forward(leaf, x, heads, **kw)
    loss = cross_entropy(logits, y)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return logits.detach(), loss.detach(), grads


def init_params(image_size, patch_size, embed_dim, num_classes, depth, heads, mlp_dim,
                channels=3, dim_head=8, masked=False, seed=0, dtype=torch.float32):
    """Random parameters with the reference's shapes (mymodel/ViT.py:125-149,
    ViT_mask.py:136-139).  Distributions are plausible stand-ins (unit-normal
    tables, 1/sqrt(fan_in)-uniform weights); parity tests load the reference's own
    tensors from fixtures instead of relying on this.
    """
    g = torch.Generator().manual_seed(seed)
    n_tok = (image_size // patch_size) ** 2 + 1
    inner = heads * dim_head

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)

    p = {}
    p["pos_embedding"] = (torch.randn((1, n_tok + 1, embed_dim) if masked else (n_tok, embed_dim),
                                      generator=g, dtype=torch.float64)).to(dtype)
    p["cls_token"] = torch.randn((1, 1, embed_dim), generator=g, dtype=torch.float64).to(dtype)
    kk = channels * patch_size * patch_size
    p["to_patch_embedding.proj.weight"] = uni((embed_dim, channels, patch_size, patch_size), kk)
    p["to_patch_embedding.proj.bias"] = uni((embed_dim,), kk)
    project_out = not (heads == 1 and dim_head == embed_dim)
    for i in range(depth):
        pre = f"transformer.layers.{i}."
        p[pre + "0.norm.weight"] = torch.ones(embed_dim, dtype=dtype)
        p[pre + "0.norm.bias"] = torch.zeros(embed_dim, dtype=dtype)
        p[pre + "0.fn.to_qkv.weight"] = uni((3 * inner, embed_dim), embed_dim)
        if project_out:
            p[pre + "0.fn.to_out.0.weight"] = uni((embed_dim, inner), inner)
            p[pre + "0.fn.to_out.0.bias"] = uni((embed_dim,), inner)
        p[pre + "1.norm.weight"] = torch.ones(embed_dim, dtype=dtype)
        p[pre + "1.norm.bias"] = torch.zeros(embed_dim, dtype=dtype)
        p[pre + "1.fn.net.0.weight"] = uni((mlp_dim, embed_dim), embed_dim)
        p[pre + "1.fn.net.0.bias"] = uni((mlp_dim,), embed_dim)
        p[pre + "1.fn.net.3.weight"] = uni((embed_dim, mlp_dim), mlp_dim)
        p[pre + "1.fn.net.3.bias"] = uni((embed_dim,), mlp_dim)
    p["mlp_head.0.weight"] = torch.ones(embed_dim, dtype=dtype)
    p["mlp_head.0.bias"] = torch.zeros(embed_dim, dtype=dtype)
    p["mlp_head.1.weight"] = uni((num_classes, embed_dim), embed_dim)
    p["mlp_head.1.bias"] = uni((num_classes,), embed_dim)
    return p
