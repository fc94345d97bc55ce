#!/bin/bash
# round 6: cache policy (nt = 2) of the role pipelines' streaming accesses: H_new stores (o), the H-step loaders' V / H loads (i), the W-statistics loaders' loads (w)
mkdir -p gpurun_out; : > gpurun_out/r6y.log
for v in prod aux_o2 aux_o2s aux_o18 aux_o16 aux_o3 prod aux_o2; do
  lib=scripts/prof_build/libsnmf_$v.so; [ $v = prod ] && lib=se_snmf_nat_amd/libsnmf_hip.so
  SNMF_LIB_PATH=$lib timeout -k 10 200 python scripts/bench_f513.py c2 a11 c4w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['shape'], round(d['iterations_per_s']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" >> gpurun_out/r6y.log || exit 1
done
cat gpurun_out/r6y.log
