#!/usr/bin/env python3
"""Times the nt GEMM forms of one ViT-B/16 encoder layer (batch 256) through the library named by VITMI_LIB, one line
per form: plain, + residual, the GELU pair, the derivative-times-gradient form with column sums."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402

DEV = "cuda:0"
T = 256 * 197
bf = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    out = []
    for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
        x = torch.randn(T, K, device=DEV).to(bf)
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(bf)
        b = torch.randn(N, device=DEV)
        r = torch.randn(T, N, device=DEV).to(bf)
        dy = torch.randn(T, N, device=DEV).to(bf)
        wt = w.t().contiguous()
        if name == "fc1":
            out.append((f"{name} gelu pair", timeit(lambda: ops.linear_fwd(x, w, b, epi=ops.EPI_GELU_D, want_aux=True))))
        elif name in ("out", "fc2"):
            out.append((f"{name} +res", timeit(lambda: ops.linear_fwd(x, w, b, res=r))))
        else:
            out.append((f"{name} plain", timeit(lambda: ops.linear_fwd(x, w, b))))
        if name == "fc2":
            aux = torch.rand(T, K, device=DEV).to(bf)
            out.append((f"{name} dx mulaux+colsum", timeit(lambda: ops.linear_bwd_x(dy, w, epi=ops.EPI_MULAUX, aux=aux, w_t=wt, want_colsum=True))))
        else:
            out.append((f"{name} dx", timeit(lambda: ops.linear_bwd_x(dy, w, w_t=wt))))
    print(os.path.basename(os.environ.get("VITMI_LIB", "default")), " ".join(f"{n}: {us:.1f}" for n, us in out))


if __name__ == "__main__":
    main()
