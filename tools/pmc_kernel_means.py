#!/usr/bin/env python3
"""Mean of every counter per kernel from rocprofv3 --pmc counter_collection.csv files.  usage: pmc_kernel_means.py <dir> [name substring ...]"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(k[:110])
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
