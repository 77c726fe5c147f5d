#!/usr/bin/env python3
"""Times the nt GEMM forms of one encoder layer at 50432 rows under every tile configuration of the library
(vitmi_debug_set_nt_config: 0 = 128x128, 2 = 256x256, 3 = 256x256 persistent), one line per configuration, with the
bytes each form has to move so that the rate can be read against the memory system.  Argument: width (default 384)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import _lib, ops  # noqa: E402

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
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    shapes = [("qkv", 3 * D, D), ("out", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)]
    forms = []
    for name, N, K in shapes:
        x = torch.randn(T, K, device=DEV).to(bf)
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(bf)
        b = torch.randn(N, device=DEV)
        r = torch.randn(T, N, device=DEV).to(bf)
        dy = torch.randn(T, N, device=DEV).to(bf)
        wt = w.t().contiguous()
        mb_f = (T * K + T * N) * 2 / 1e6
        if name == "fc1":
            forms.append((f"{name} gelu pair", mb_f + T * N * 2 / 1e6,
                          lambda x=x, w=w, b=b: ops.linear_fwd(x, w, b, epi=ops.EPI_GELU_D, want_aux=True)))
        elif name in ("out", "fc2"):
            forms.append((f"{name} +res", mb_f + T * N * 2 / 1e6, lambda x=x, w=w, b=b, r=r: ops.linear_fwd(x, w, b, res=r)))
        else:
            forms.append((f"{name} plain", mb_f, lambda x=x, w=w, b=b: ops.linear_fwd(x, w, b)))
        if name == "fc2":
            aux = torch.rand(T, K, device=DEV).to(bf)
            forms.append((f"{name} dx mulaux+colsum", mb_f + T * K * 2 / 1e6,
                          lambda dy=dy, w=w, wt=wt, aux=aux: ops.linear_bwd_x(dy, w, epi=ops.EPI_MULAUX, aux=aux, w_t=wt,
                                                                              want_colsum=True)))
        else:
            forms.append((f"{name} dx", mb_f, lambda dy=dy, w=w, wt=wt: ops.linear_bwd_x(dy, w, w_t=wt)))
    print("MB moved:", " ".join(f"{n}: {mb:.0f}" for n, mb, _ in forms))
    for cfg in (-1, 0, 2, 3):
        _lib.check(_lib.lib.vitmi_debug_set_nt_config(cfg))
        print(f"cfg {cfg:2d}", " ".join(f"{n}: {timeit(fn):.1f}" for n, _, fn in forms), flush=True)
    _lib.check(_lib.lib.vitmi_debug_set_nt_config(-1))


if __name__ == "__main__":
    main()
