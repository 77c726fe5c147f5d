#!/usr/bin/env python3
"""How much of the thin last round of an input-gradient product can hide under the LayerNorm backward that follows
it: the product's tail rows on a second stream beside LayerNorm backward over the head rows, against the two run one
after the other.  Shapes of the ViT-B/16 step at batch 256 (50432 tokens)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402

M, D, M1 = 50432, 768, 170 * 256
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
dres = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
gamma = torch.ones(D, device=dev)
mean = x.float().mean(1).contiguous()
rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
side = torch.cuda.Stream()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K in (3072, 2304, 768):
    du = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(K, D, device=dev, generator=g) * 0.03).to(torch.bfloat16)      # y = x W^T with W [K(out), D(in)]
    wt = w.t().contiguous()

    def seq():
        dxn = ops.linear_bwd_x(du, w, w_t=wt)
        return ops.layernorm_bwd(dxn, x, gamma, mean, rstd, dres=dres)

    def ovl():
        main = torch.cuda.current_stream()
        dxn_h = ops.linear_bwd_x(du[:M1], w, w_t=wt)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dxn_t = ops.linear_bwd_x(du[M1:], w, w_t=wt)
        out_h = ops.layernorm_bwd(dxn_h, x[:M1], gamma, mean[:M1], rstd[:M1], dres=dres[:M1])
        main.wait_stream(side)
        out_t = ops.layernorm_bwd(dxn_t, x[M1:], gamma, mean[M1:], rstd[M1:], dres=dres[M1:])
        return out_h, out_t

    def gemm_only():
        return ops.linear_bwd_x(du, w, w_t=wt)

    def ln_only(dxn=torch.empty(M, D, dtype=torch.bfloat16, device=dev)):
        return ops.layernorm_bwd(dxn, x, gamma, mean, rstd, dres=dres)

    # replayed from HIP graphs, so that the host's launch rate is out of the picture
    graphs = {}
    for name, fn in (("seq", seq), ("ovl", ovl)):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            keep = fn()
        graphs[name] = (gr, keep)
    ts = {"seq": [], "ovl": []}
    for rep in range(3):
        for name in ts:
            ts[name].append(timed(graphs[name][0].replay))
    print(f"K {K}: product {timed(gemm_only):.1f} us, LayerNorm backward {timed(ln_only):.1f} us, one after the other "
          f"{min(ts['seq']):.1f} us, tail beside the head rows' LayerNorm {min(ts['ovl']):.1f} us", flush=True)
