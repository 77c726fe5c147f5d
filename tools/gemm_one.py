#!/usr/bin/env python3
"""Runs a handful of launches of one GEMM shape (for rocprofv3 --pmc passes).
usage: gemm_one.py {nt|tn} N K [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402

kind, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
T = 256 * 197
bf = torch.bfloat16
x = torch.randn(T, K, device="cuda").to(bf)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
dy = torch.randn(T, N, device="cuda").to(bf)
for _ in range(iters):
    if kind == "nt":
        ops.linear_fwd(x, w)
    else:
        ops.linear_bwd_w(dy, x)
torch.cuda.synchronize()
