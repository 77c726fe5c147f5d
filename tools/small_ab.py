#!/usr/bin/env python3
"""Times the block calls at the small geometry with the one-launch kernels on and off (same box, same process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd._lib import check, lib  # noqa: E402
from visiontransformer_amd.ops import _p, _stream  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16


def timeit(fn, per_graph=20, replays=20):
    """Device time per call in us: the calls are captured into one HIP graph and the graph is replayed, so the host's
    cost per foreign call (about 14 us through ctypes) stays out of the figure."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (replays * per_graph) * 1e3


def ff(rows, D, M):
    x = torch.randn(rows, D, device=DEV).to(BF)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    w1 = (torch.randn(M, D, device=DEV) * D ** -0.5).to(BF)
    w2 = (torch.randn(D, M, device=DEV) * M ** -0.5).to(BF)
    b1, b2 = torch.zeros(M, device=DEV), torch.zeros(D, device=DEV)
    xn, y = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    u, h = torch.empty(rows, M, dtype=BF, device=DEV), torch.empty(rows, M, dtype=BF, device=DEV)
    def run():
        s = _stream()
        check(lib.vitmi_ff_block_fwd(1, rows, D, M, _p(x), _p(gamma), _p(beta), 1e-5, _p(w1), _p(b1), _p(w2), _p(b2),
                                     _p(xn), _p(mean), _p(rstd), _p(u), _p(h), _p(y), s))
    out = []
    for mask in (0, 1):
        check(lib.vitmi_debug_set_small_fused(mask))
        out.append(timeit(run))
    print(f"ff fwd rows {rows} D {D} M {M}: steps {out[0]:.1f} us, one launch {out[1]:.1f} us")
    if os.environ.get("SMALL_ABLATE"):
        for mt in (2, 3, 4, 5, 6):
            check(lib.vitmi_debug_set_small_fused(1 | (mt << 16)))
            print(f"   {mt} token tiles per workgroup: {timeit(run):.1f} us")
    check(lib.vitmi_debug_set_small_fused(-1))


def attn(B, N, D, H, DH):
    I = H * DH
    x = torch.randn(B * N, D, device=DEV).to(BF)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    wq = (torch.randn(3 * I, D, device=DEV) * D ** -0.5).to(BF)
    wo = (torch.randn(D, I, device=DEV) * I ** -0.5).to(BF)
    bo = torch.zeros(D, device=DEV)
    xn, y = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(B * N, device=DEV), torch.empty(B * N, device=DEV)
    qkv = torch.empty(B * N, 3 * I, dtype=BF, device=DEV)
    o = torch.empty(B * N, I, dtype=BF, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)

    def run():
        check(lib.vitmi_attn_block_fwd(1, B, N, D, H, DH, _p(x), _p(gamma), _p(beta), 1e-5, _p(wq), None, _p(wo), _p(bo),
                                       _p(xn), _p(mean), _p(rstd), _p(qkv), _p(o), _p(lse), None, _p(y), 0, _stream()))
    out = []
    for mask in (0, 3):
        check(lib.vitmi_debug_set_small_fused(mask))
        out.append(timeit(run))
    check(lib.vitmi_debug_set_small_fused(-1))
    print(f"attention fwd B {B} N {N} D {D} H {H} dh {DH}: steps {out[0]:.1f} us, one launch {out[1]:.1f} us")


if __name__ == "__main__":
    for B in (256, 64):
        ff(B * 65, 192, 512)
        attn(B, 65, 192, 12, 8)
