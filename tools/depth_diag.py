#!/usr/bin/env python3
"""Per-tensor gradient error (maximum and rms measure) of the full-depth ViT-B/16 in bf16 against the CPU oracle, for
several batch sizes and seeds."""
import os
import sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import vit_oracle
from visiontransformer_amd import ViT
kw = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072, dim_head=64)
for B, seed in ((8, 9), (8, 10), (32, 9)):
    params = vit_oracle.init_params(masked=False, seed=seed, **kw)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, 3, 224, 224, generator=g); y = torch.randint(0, 100, (B,), generator=g)
    logits_ref, loss_ref, grads_ref = vit_oracle.loss_and_grads(params, x, y, 12)
    m = ViT.VIT(**kw); m.load_state_dict(params); m = m.to('cuda').set_compute_dtype(torch.bfloat16).train()
    logits = m(x.cuda()); loss = torch.nn.functional.cross_entropy(logits, y.cuda()); loss.backward()
    rows = []
    for k, p in m.named_parameters():
        a = p.grad.double().cpu(); b = grads_ref[k].double()
        rows.append(((a - b).abs().max().item() / b.abs().max().item(), ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item(), k))
    rows.sort(reverse=True)
    print(B, seed, 'logits', ((logits.double().cpu() - logits_ref.double()).abs().max() / logits_ref.abs().max()).item())
    for r in rows[:4]: print('   max-rel %.4f rms-rel %.4f %s' % r)
    print('   worst rms-rel %.4f' % max(r[1] for r in rows))
