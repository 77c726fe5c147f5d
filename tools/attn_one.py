#!/usr/bin/env python3
"""A few launches of the attention kernels on the ViT-B/16 shape, for profiler runs.  usage: attn_one.py [stream 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402

B, N, H, DH = 256, 197, 12, 64
lib.vitmi_debug_set_attn_stream(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
for _ in range(5):
    o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)
torch.cuda.synchronize()
