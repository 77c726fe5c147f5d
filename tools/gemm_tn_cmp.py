#!/usr/bin/env python3
"""Compare tn (weight-gradient) tile configurations (vitmi_debug_set_tn_config) on the ViT-B/16
shapes: checks each against a torch matmul and prints its rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd._lib import BF16, check, lib  # noqa: E402

T = 256 * 197
bf = torch.bfloat16
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


s = torch.cuda.current_stream().cuda_stream
for N, K in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn(T, K, device="cuda").to(bf)
    dy = torch.randn(T, N, device="cuda").to(bf)
    ref = dy[:, :256].float().t() @ x.float()
    fl = 2.0 * T * N * K
    for cfg in cfgs:
        lib.vitmi_debug_set_tn_config(cfg)
        dw = torch.zeros(N, K, device="cuda", dtype=torch.float32)
        ws = torch.empty(max(lib.vitmi_gemm_wgrad_workspace_bytes(BF16, N, K, T), 16), dtype=torch.uint8, device="cuda")
        run = lambda: check(lib.vitmi_gemm_wgrad(BF16, N, K, T, dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), K,  # noqa: E731
                                                ws.data_ptr(), s))
        run()
        err = ((dw[:256] - ref).abs().max() / ref.abs().max()).item()
        full = (dw - dy.float().t() @ x.float()).abs().max().item() / ref.abs().max().item()
        ms = timeit(run)
        ms_al = timeit(lambda: check(lib.vitmi_gemm_wgrad(BF16, N, K, T, dy.data_ptr(), 0, x.data_ptr(), 0, dw.data_ptr(), K,
                                                         ws.data_ptr(), s)))
        print(f"tn {N}x{K}x{T} cfg {cfg}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF  rel err {err:.2e} full {full:.2e}"
              f" | aliased rows {fl / ms_al / 1e9:7.1f} TF", flush=True)
lib.vitmi_debug_set_tn_config(-1)
