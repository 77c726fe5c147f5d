#!/usr/bin/env python3
"""Target of a rocprofv3 --pmc FETCH_SIZE run: the plain nt GEMM at 50432 x N x K, two launches per tile-walk group
size in the order given, so the counter rows (in dispatch order) read as one pair per group size.
usage: gemm_group_pmc.py N K gm,gm,...   |   gemm_group_pmc.py --read <dir> gm,gm,..."""
import csv
import glob
import os
import sys

if sys.argv[1] == "--read":
    gms = sys.argv[3].split(",")
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt_phased" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals = [v for _, v in rows][1:]          # the first launch is a warm-up
    for i, gm in enumerate(gms):
        pair = vals[2 * i:2 * i + 2]
        print(f"gm {gm:>3s}: fetch " + " ".join(f"{2 * v * 1024 / 1e6:7.1f}" for v in pair) + " MB per launch")
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from visiontransformer_amd import ops  # noqa: E402
from visiontransformer_amd._lib import BF16, lib  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
T = 256 * 197
bf = torch.bfloat16
x = torch.randn(T, K, device="cuda").to(bf)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
y = torch.empty(T, N, device="cuda", dtype=bf)
ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16)
for gm in sys.argv[3].split(","):
    lib.vitmi_debug_set_nt_group(int(gm))
    for _ in range(2):
        ops.gemm_raw(BF16, T, N, K, x, K, 1, w, K, 1, y, N, BF16)
torch.cuda.synchronize()
print("done")
