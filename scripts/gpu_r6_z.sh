#!/bin/bash
# round 6: the H step's loads of H_in (dead after the launch) as non-temporal loads, on top of the non-temporal H_new stores
mkdir -p gpurun_out; : > gpurun_out/r6z.log
for v in prod aux_wh2 aux_wv2 prod aux_wh2 aux_wv2; do
  lib=scripts/prof_build/libsnmf_$v.so; [ $v = prod ] && lib=se_snmf_nat_amd/libsnmf_hip.so
  SNMF_LIB_PATH=$lib timeout -k 10 200 python scripts/bench_f513.py c2 a11 c4w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['shape'], round(d['iterations_per_s'],1), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" >> gpurun_out/r6z.log || exit 1
done
cat gpurun_out/r6z.log
