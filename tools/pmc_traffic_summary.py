#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, each its own run with
--output-format csv): bytes per launch = FETCH_SIZE KB x 1024 x 2 (gfx950 tallies a 128-byte read request as 64
bytes) + WRITE_SIZE KB x 1024.
usage: pmc_traffic_summary.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run> out.json "<what was run>" """
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(root, counter):
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {root}")
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_hash  # noqa: E402  (bench.py refuses a summary collected on other kernel sources)

out = {"kernel_source_hash": kernel_source_hash(),
       "_about": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `{sys.argv[4]}`, MI355X. Counter unit "
                 "is KB per dispatch; bytes = KB*1024; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B).",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
    if k not in write or nf[k] != nw[k]:
        continue
    n = nf[k]
    out["kernels"][k] = {"launches": n, "fetch_size_kb_raw": round(fetch[k] / n, 1), "write_size_kb": round(write[k] / n, 1),
                         "hbm_bytes_per_launch": int((2 * fetch[k] + write[k]) / n * 1024)}
with open(sys.argv[3], "w") as f:
    json.dump(out, f, indent=1)
for k, v in list(out["kernels"].items())[:14]:
    print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch x{v['launches']:4d}  {k[:110]}")
