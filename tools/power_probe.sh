#!/bin/bash
# Samples the board power, the power cap and the shader clock (rocm-smi, read-only) while bench.py runs the headline
# workload; prints min / mean / max of the samples taken inside the timed steps.  usage: tools/power_probe.sh [steps]
steps=${1:-300}
out=${2:-gpurun_out/power_probe.txt}
mkdir -p "$(dirname "$out")"
python3 bench.py --steps "$steps" --warmup 10 --no-cpu-baseline > "$out.bench" 2>/dev/null &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> "$out"
  echo "---" >> "$out"
  sleep 0.4
done
wait $pid
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read().split('---')
pw, sc, cap = [], [], []
for blk in txt:
    m = re.search(r'Average Graphics Package Power \(W\): ([0-9.]+)|Current Socket Graphics Package Power \(W\): ([0-9.]+)', blk)
    if m: pw.append(float(m.group(1) or m.group(2)))
    m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', blk)
    if m: sc.append(int(m.group(1)))
    m = re.search(r'Max Graphics Package Power \(W\): ([0-9.]+)', blk)
    if m: cap.append(float(m.group(1)))
def st(v): return (min(v), sum(v) / len(v), max(v)) if v else None
print("samples", len(txt), "power W (min, mean, max)", st(pw), "cap", st(cap), "sclk MHz", st(sc))
# the busy part: samples above 60 % of the maximum
if pw:
    busy = [p for p in pw if p > 0.6 * max(pw)]
    print("busy samples", len(busy), "mean power W", sum(busy) / len(busy))
PY
tail -1 "$out.bench" | cut -c1-200
