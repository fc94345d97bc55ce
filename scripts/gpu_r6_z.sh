#!/bin/bash
# round 6: non-temporal H_new stores in the un-pipelined H step too (stage_out)?  C5 and the shapes that run k_hstep
mkdir -p gpurun_out; : > gpurun_out/r6z.log
for v in prod aux_sont prod aux_sont; do
  lib=scripts/prof_build/libsnmf_$v.so; [ $v = prod ] && lib=se_snmf_nat_amd/libsnmf_hip.so
  SNMF_LIB_PATH=$lib timeout -k 10 200 python scripts/bench_f513.py c5 c2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['shape'], round(d['iterations_per_s'],1), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" >> gpurun_out/r6z.log || exit 1
done
cat gpurun_out/r6z.log
