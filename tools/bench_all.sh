#!/bin/bash
# usage: tools/bench_all.sh <tag>: one bench.py line per workload (with the CPU baseline) into gpurun_out/<tag>_w_<workload>.json
set -o pipefail
tag=$1
cd "$(dirname "$0")/.." || exit 1
for w in vit_b16_224 vit_b16_224_f32 cifar10_cpu cifar100_export cifar_mask cifar_mask_pbf dpsgd_vit_b16 tim_vit_small_mask; do
    timeout -k 10 400 python3 bench.py --workload $w > gpurun_out/${tag}_w_$w.log 2>&1 || { tail -20 gpurun_out/${tag}_w_$w.log; exit 1; }
    grep '^{' gpurun_out/${tag}_w_$w.log | tail -1 > gpurun_out/${tag}_w_$w.json
    python3 -c "import json,sys; d=json.load(open('gpurun_out/${tag}_w_$w.json')); print('$w', round(d['value']), d['unit'], round(d['ms_per_step'],2), 'ms', '| roofline', d['roofline']['kernel'][:30], round(d['roofline']['frac'],3), '| cpu', round(d['cpu_baseline']['value'],1))"
done
