#!/bin/bash
# usage: stepab.sh <variant> ... ; alternates base and variants, 2 rounds, prints ms/step and family shares
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['time_share_ms_per_step']
g=lambda k: [v for kk,v in t.items() if k in kk][0]
print('$1', round(d['ms_per_step'],3), 'gelu', g('[gelu->bf16]'), 'dgelu', g('[dgelu->bf16]'), 'plain', g('[plain->bf16]'), 'tn', g('gemm_tn'), 'ln', g('layernorm'), 'attn', t['attn_fwd'], t['attn_bwd'])"; }
for i in 1 2; do
  unset VITMI_LIB; timeout -k 10 100 python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | show base
  for v in "$@"; do VITMI_LIB=$PWD/tools/altlib/libvitmi_hip_$v.so timeout -k 10 100 python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | show $v; done
done
