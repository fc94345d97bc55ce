#!/bin/bash
# as phase_prof.sh, raw: every [SNMF_PROF] line and the kernel times of each variant given as an argument
cd "$(dirname "$0")/.."
export SNMF_LIB_PATH="$PWD/scripts/prof_build/libsnmf_hip_prof.so"
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps ${STEPS:-200} --warmup 2 --no-cpu-baseline > /tmp/pp.out 2> /tmp/pp.err
  grep "SNMF_PROF" /tmp/pp.err | tail -2
  python -c "
import json
d=json.loads(open('/tmp/pp.out').read().strip().splitlines()[-1]); print(round(d['value'],1), {k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()})"
done
