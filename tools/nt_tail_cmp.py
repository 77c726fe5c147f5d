#!/usr/bin/env python3
"""fc2's input-gradient product (50432 x 3072 x 768, times the stored derivative) with and without the rows of the
thin last round on the 128 x 128 kernel, with and without the fused column sums; alternating in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402


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


M, N, K = 50432, 768, 3072           # dy [M, N] @ w [N, K] -> [M, K]
g = torch.Generator(device="cuda").manual_seed(5)
dy = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
wt = w.t().contiguous()
aux = torch.rand(M, K, device="cuda", generator=g).half().view(torch.bfloat16)
for cs in (False, True):
    res = {0: [], 1: []}
    for rep in range(3):
        for on in (1, 0):
            lib.vitmi_debug_set_nt_tail_split(on)
            res[on].append(timed(lambda: ops.linear_bwd_x(dy, w, epi=ops.EPI_MULAUX, aux=aux, w_t=wt, want_colsum=cs)))
    lib.vitmi_debug_set_nt_tail_split(1)
    print(f"column sums {cs}: tail split on {min(res[1]):.1f} us, off {min(res[0]):.1f} us", flush=True)
