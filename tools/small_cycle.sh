#!/bin/bash
# tests of the one-launch small-geometry kernels, then a kernel trace of tools/small_ab.py (average durations per kernel)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
root=$(pwd)
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_small_fused.py -x -q --timeout 120 --timeout-method thread > gpurun_out/sm_test.log 2>&1 || { tail -30 gpurun_out/sm_test.log; exit 1; }
tail -1 gpurun_out/sm_test.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof_sm
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_sm" -o sm -- python3 "$root/tools/small_ab.py" > "$root/gpurun_out/sm_ab.log" 2>&1 || { tail -20 "$root/gpurun_out/sm_ab.log"; exit 1; }
cd "$root"
find gpurun_out/prof_sm -name "*kernel_trace.csv" -delete
grep -v amdgpu.ids gpurun_out/sm_ab.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_sm/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    if "at::native" in r["Name"]:
        continue
    print(f'{float(r["AverageNs"]) / 1e3:8.1f} us x {r["Calls"]:>5}  {r["Name"][:110]}')
PY
