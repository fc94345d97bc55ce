#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export SNMF_LIB_PATH=scripts/prof_build/libsnmf_ovexp.so
for ov in 0 1; do for sp in 1 0; do
  SNMF_OVERLAP=$ov SNMF_HSTEP_SPLIT=$sp timeout -k 10 200 python scripts/bench_f513.py c2 a11 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('overlap=$ov split=$sp', d['shape'], round(d['iterations_per_s'],1), round(d['ms_per_iteration']*1e3,1), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})
"
done; done | tee gpurun_out/r6i_overlap.log
