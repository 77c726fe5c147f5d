#!/bin/bash
# Kernel trace of the cifar_mask bench workload: writes gpurun_out/prof_cifar/ and prints the top kernels.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_cifar -o cifar -- python3 $root/bench.py --workload cifar_mask --no-cpu-baseline --steps 30 --warmup 3 > $root/gpurun_out/prof_cifar.log 2>&1 || { tail $root/gpurun_out/prof_cifar.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$root/gpurun_out/prof_cifar/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("total ms", sum(float(r['TotalDurationNs']) for r in rows) / 1e6)
for r in rows[:8]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.1f}")
PY
tail -1 $root/gpurun_out/prof_cifar.log | cut -c1-200
