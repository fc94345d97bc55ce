#!/bin/bash
# diagnostic (PROF library): operand-reuse experiments of k_hstep_rp; prints kernel times and the final cost (a changed
# cost proves the experiment really altered the operand streams)
cd "$(dirname "$0")/.."
export SNMF_LIB_PATH="$PWD/scripts/prof_build/libsnmf_hip_prof.so"
for x in ${XSELS:--1 6 7 8}; do
  SNMF_HSTEP_DUO=${DUO:-0} SNMF_STAGGER=1,0 SNMF_STAGGER_SHIFT=$x python bench.py --steps 60 --warmup 2 --no-cpu-baseline > /tmp/x.out 2>/tmp/x.err
  python -c "
import json
d=json.loads(open('/tmp/x.out').read().strip().splitlines()[-1]); print('xsel=$x', {k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()}, 'final_cost', d['final_cost'])"
  grep -o "shares.*stage_out=[0-9.]*%" /tmp/x.err | tail -1
done
