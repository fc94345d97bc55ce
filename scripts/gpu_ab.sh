#!/bin/bash
# quick A/B of env variants on the product library: scripts/gpu_ab.sh "VAR=1" "VAR=0 OTHER=2" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  env $v python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > /tmp/ab.out 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
  python -c "
import json
d=json.loads(open('/tmp/ab.out').read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'it/s', {k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()}, 'cost_vs_oracle', (d.get('cost_vs_oracle') or {}).get('rel_diff'))"
done
