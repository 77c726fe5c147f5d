#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace) into the per-kernel statistics table that
`--stats` prints: calls, total / average / min / max duration (ns) and share of GPU time.
usage: rocpd_kernel_stats.py results.db out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
    "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, n, tot, avg, mn, mx in rows:
        w.writerow([name, n, tot, f"{avg:.1f}", f"{100.0 * tot / total:.2f}", mn, mx])
print(f"{len(rows)} kernels, {total / 1e6:.2f} ms of kernel time -> {sys.argv[2]}")
