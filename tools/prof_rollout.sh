#!/bin/bash
# Kernel trace of tools/rollout_phase.py: writes gpurun_out/prof_rollout/ and prints the kernels.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_rollout -o roll -- python3 $root/tools/rollout_phase.py > $root/gpurun_out/prof_rollout.log 2>&1 || { tail $root/gpurun_out/prof_rollout.log; exit 1; }
cat $root/gpurun_out/prof_rollout.log | grep discard
python3 - <<PY
import csv, glob
f = glob.glob('$root/gpurun_out/prof_rollout/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.1f}")
PY
