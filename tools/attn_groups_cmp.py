#!/usr/bin/env python3
"""Short sequences of narrow heads: six heads of an image per workgroup against one (image, head) pair per workgroup,
one process, one box.  Outputs must be bit-identical; the two are timed alternately."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import lib  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B, N, H, DH in [(256, 65, 12, 8), (256, 65, 12, 16), (256, 17, 12, 8), (64, 65, 12, 8), (256, 65, 6, 8)]:
    torch.manual_seed(N + DH)
    qkv = torch.randn(B, N, 3 * H * DH, device="cuda").to(torch.bfloat16)
    do = torch.randn(B, N, H * DH, device="cuda").to(torch.bfloat16)
    res = {}
    for on in (0, 1):
        lib.vitmi_debug_set_attn_head_groups(on)
        o, lse, _ = ops.attn_fwd(qkv, B, N, H, DH, False)
        o2, lse2, pr = ops.attn_fwd(qkv, B, N, H, DH, True)
        dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)
        torch.cuda.synchronize()
        res[on] = (o.clone(), lse.clone(), o2.clone(), pr.clone(), dqkv.clone())
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    t = {0: [[], [], []], 1: [[], [], []]}
    for rep in range(3):
        for on in (0, 1):
            lib.vitmi_debug_set_attn_head_groups(on)
            t[on][0].append(timed(lambda: ops.attn_fwd(qkv, B, N, H, DH, False)))
            t[on][1].append(timed(lambda: ops.attn_fwd(qkv, B, N, H, DH, True)))
            t[on][2].append(timed(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, DH)))
    print(f"B {B} N {N} H {H} dh {DH}: identical {same}; per pair / six heads per workgroup: fwd {min(t[0][0]):.1f} / "
          f"{min(t[1][0]):.1f} us, fwd+export {min(t[0][1]):.1f} / {min(t[1][1]):.1f} us, bwd {min(t[0][2]):.1f} / "
          f"{min(t[1][2]):.1f} us", flush=True)
lib.vitmi_debug_set_attn_head_groups(1)
