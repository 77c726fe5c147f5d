#!/usr/bin/env python3
"""Mean of every SQ counter per attention kernel from two rocprofv3 --pmc passes (8 counters each) over
tools/attn_one.py.  usage: pmc_attention_summary.py <dir of pass 1> <dir of pass 2> out.json "<about>" """
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in sys.argv[1:3]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_about": sys.argv[4], "kernels": {}}
for k, cs in sorted(acc.items()):
    if "attn_" not in k:
        continue
    import re
    short = re.search(r"(attn_[A-Za-z0-9_]*?kernel)", k).group(1)
    d = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    if d.get("SQ_WAVE_CYCLES"):
        d["WAIT_ANY_over_WAVE_CYCLES"] = round(d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"], 3)
        d["WAIT_INST_ANY_over_WAVE_CYCLES"] = round(d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"], 3)
    out["kernels"][short] = d
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
