#!/bin/bash
# round 6: start stagger of the SIMD partners in the small-F kernels (k_hstep_sf: SNMF_SF_STAG, k_wstats_sf: SNMF_WSF_STAG), cycles
mkdir -p gpurun_out; : > gpurun_out/r6s.log
for st in 0 4000 8000 16000 30000; do
  SNMF_SF_STAG=$st timeout -k 10 100 python scripts/bench_f513.py melh 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('SNMF_SF_STAG=$st', d['shape'], round(d['iterations_per_s']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" >> gpurun_out/r6s.log || exit 1
  SNMF_WSF_STAG=$st timeout -k 10 100 python scripts/bench_f513.py melw 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('SNMF_WSF_STAG=$st', d['shape'], round(d['iterations_per_s']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" >> gpurun_out/r6s.log || exit 1
done
cat gpurun_out/r6s.log
