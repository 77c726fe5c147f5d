#!/usr/bin/env python3
"""One GEMM shape, a few launches: the target of rocprofv3 --pmc runs.
usage: gemm_one.py nt|tn cfg N K [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, check, lib  # noqa: E402

kind, cfg, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
T = 256 * 197
bf = torch.bfloat16
x = torch.randn(T, K, device="cuda").to(bf)
s = torch.cuda.current_stream().cuda_stream
if kind == "nt":
    lib.vitmi_debug_set_nt_config(cfg)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
    y = torch.empty(T, N, device="cuda", dtype=bf)
    for _ in range(iters):
        ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16)
else:
    lib.vitmi_debug_set_tn_config(cfg)
    dy = torch.randn(T, N, device="cuda").to(bf)
    dw = torch.empty(N, K, device="cuda", dtype=torch.float32)
    ws = torch.empty(max(lib.vitmi_gemm_wgrad_workspace_bytes(BF16, N, K, T), 16), dtype=torch.uint8, device="cuda")
    for _ in range(iters):
        check(lib.vitmi_gemm_wgrad(BF16, N, K, T, dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), K, ws.data_ptr(), s))
torch.cuda.synchronize()
print("done")
