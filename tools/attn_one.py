#!/usr/bin/env python3
"""A few launches of the attention kernels, for profiler runs.  usage: attn_one.py [B N H DH] (default: the ViT-B/16 shape)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402

B, N, H, DH = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 197, 12, 64)
# heads of 64: qkv as the head-major planes the training step keeps them in (the same number of elements)
layout = ops.QKV_HEAD_MAJOR if DH == 64 and "rows" not in sys.argv else 0
qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
for _ in range(5):
    o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False, layout)
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH, layout)
torch.cuda.synchronize()
