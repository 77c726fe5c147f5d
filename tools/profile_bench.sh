#!/bin/bash
# usage: tools/profile_bench.sh <tag> [bench.py arguments ...]
# rocprofv3 --kernel-trace --stats of `python3 bench.py <args>`; leaves gpurun_out/<tag>_kernel_stats.csv (per-kernel calls,
# total / average duration) and gpurun_out/<tag>_bench.log (the bench line printed under the profiler).
set -o pipefail
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 bench.py "$@" > gpurun_out/${tag}_bench.log 2>&1 || { tail -20 gpurun_out/${tag}_bench.log; exit 1; }
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv under $out"; find "$out" | head; exit 1; }
cp "$f" gpurun_out/${tag}_kernel_stats.csv
find "$out" -name "*kernel_trace.csv" -delete
head -25 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
grep '^{' gpurun_out/${tag}_bench.log | tail -1 | cut -c1-300
