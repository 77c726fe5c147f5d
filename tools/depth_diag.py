#!/usr/bin/env python3
"""Per-tensor gradient error (maximum and rms measure) of the full-depth ViT-B/16 in bf16 against the CPU oracle, for
several batch sizes and seeds, with both GELU modes, beside a yardstick: the same oracle run by torch on the same GPU
under bf16 autocast (fp32 master weights, bf16 matrix products, fp32 LayerNorm / softmax), i.e. what the dtype itself
costs on these tensors."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import vit_oracle  # noqa: E402
import visiontransformer_amd as vta  # noqa: E402
from visiontransformer_amd import ViT  # noqa: E402

kw = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072, dim_head=64)


def errs(grads, ref):
    rows = []
    for k, g in grads.items():
        a, b = g.double().cpu(), ref[k].double()
        rows.append(((a - b).abs().max().item() / b.abs().max().item(),
                     ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item(), k))
    rows.sort(reverse=True)
    return rows


for B, seed in ((8, 9), (8, 10), (32, 9)):
    params = vit_oracle.init_params(masked=False, seed=seed, **kw)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 100, (B,), generator=g)
    logits_ref, loss_ref, grads_ref = vit_oracle.loss_and_grads(params, x, y, 12)
    pc = {k: v.cuda() for k, v in params.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lg, _, gy = vit_oracle.loss_and_grads(pc, x.cuda(), y.cuda(), 12)
    rows = errs(gy, grads_ref)
    print(f"B {B} seed {seed} torch-autocast yardstick: logits {((lg.double().cpu() - logits_ref.double()).abs().max() / logits_ref.abs().max()).item():.4f}")
    for r in rows[:3]:
        print('   max-rel %.4f rms-rel %.4f %s' % r)
    yard = {k: (a, b) for a, b, k in rows}
    # the all-bf16 run: parameters, residual stream and every operator's input and output in bf16 (torch's bf16
    # operators accumulate in fp32 inside one operator only), the arithmetic the HIP path's bf16 mode is closest to
    pb = {k: v.cuda().bfloat16() for k, v in params.items()}
    lgb, _, gyb = vit_oracle.loss_and_grads(pb, x.cuda().bfloat16(), y.cuda(), 12)
    rows_b = errs(gyb, grads_ref)
    print(f"   all-bf16 torch: logits {((lgb.double().cpu() - logits_ref.double()).abs().max() / logits_ref.abs().max()).item():.4f}")
    for r in rows_b[:4]:
        print('   max-rel %.4f rms-rel %.4f %s' % r)
    yard_b = {k: (a, b) for a, b, k in rows_b}
    for mode in ("fast", "erf"):
        vta.set_gelu(mode)
        m = ViT.VIT(**kw)
        m.load_state_dict(params)
        m = m.to('cuda').set_compute_dtype(torch.bfloat16).train()
        logits = m(x.cuda())
        torch.nn.functional.cross_entropy(logits, y.cuda()).backward()
        rows = errs({k: p.grad for k, p in m.named_parameters()}, grads_ref)
        over = [(a, yard[k][0], k) for a, b, k in rows if a > yard[k][0]]
        print(f"  HIP gelu={mode}: logits {((logits.double().cpu() - logits_ref.double()).abs().max() / logits_ref.abs().max()).item():.4f}; "
              f"{len(over)} of {len(rows)} tensors above the yardstick's maximum error")
        for r in rows[:4]:
            print('   max-rel %.4f rms-rel %.4f %s   (yardstick %.4f)' % (r + (yard[r[2]][0],)))
        print('   worst rms-rel %.4f' % max(r[1] for r in rows))
        over_b = [(a, yard_b[k][0], k) for a, b, k in rows if a > yard_b[k][0]]
        print(f"   {len(over_b)} tensors above the all-bf16 run's maximum error:", over_b[:6])
    vta.set_gelu("fast")
