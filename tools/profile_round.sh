#!/bin/bash
# usage: tools/profile_round.sh <tag>: kernel traces of three workloads, the two HBM-traffic counter passes over the
# headline workload and the two SQ counter passes over the attention kernels; everything under gpurun_out/<tag>_*
set -o pipefail
tag=$1
cd "$(dirname "$0")/.." || exit 1
root=$(pwd)
export TMPDIR=/tmp
tools/profile_bench.sh ${tag}_bench --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_prof1.log 2>&1 || { tail gpurun_out/${tag}_prof1.log; exit 1; }
tools/profile_bench.sh ${tag}_tim_vit_small --workload tim_vit_small_mask --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_prof2.log 2>&1 || { tail gpurun_out/${tag}_prof2.log; exit 1; }
tools/profile_bench.sh ${tag}_cifar_mask --workload cifar_mask --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_prof3.log 2>&1 || { tail gpurun_out/${tag}_prof3.log; exit 1; }
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/${tag}_pmc_$c
    ( cd /tmp && timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/${tag}_pmc_$c -o pmc -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers ) > gpurun_out/${tag}_pmc_$c.log 2>&1 || { tail gpurun_out/${tag}_pmc_$c.log; exit 1; }
done
python3 tools/pmc_traffic_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_pmc_traffic.json "$cmd" > gpurun_out/${tag}_pmc_summary.log 2>&1 || { tail gpurun_out/${tag}_pmc_summary.log; exit 1; }
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
p2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
i=1
for set in "$p1" "$p2"; do
    rm -rf gpurun_out/${tag}_pmc_attn$i
    ( cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $root/gpurun_out/${tag}_pmc_attn$i -o pmc -- python3 $root/tools/attn_one.py ) > gpurun_out/${tag}_pmc_attn$i.log 2>&1 || { tail gpurun_out/${tag}_pmc_attn$i.log; exit 1; }
    i=$((i + 1))
done
python3 tools/pmc_attention_summary.py gpurun_out/${tag}_pmc_attn1 gpurun_out/${tag}_pmc_attn2 gpurun_out/${tag}_pmc_attention.json "rocprofv3 --pmc (two passes of 8 SQ counters each) over tools/attn_one.py: bf16 attention at B 256, N 197, H 12, dh 64 with qkv in head-major planes, five launches per kernel, means per launch, summed over the chip" > /dev/null || exit 1
# the counter CSVs are large: keep the summaries only
find gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_pmc_attn1 gpurun_out/${tag}_pmc_attn2 -name "*.csv" -delete
echo done
