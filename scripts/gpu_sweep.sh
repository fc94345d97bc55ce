#!/bin/bash
# one gpurun call: bench_f513 shapes under a list of environment settings:  scripts/gpu_sweep.sh "<shapes>" "ENV=a ENV=b ..." [iters]
SHAPES=$1; SETS=$2; IT=${3:-60}
for s in $SETS; do
  echo "== $s"
  env $s timeout -k 10 300 python scripts/bench_f513.py $SHAPES --iters $IT 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['shape'], round(d['iterations_per_s'], 1), 'it/s', {k: round(v, 4) for k, v in d['kernel_ms'].items()}, {k: round(v, 3) for k, v in d['kernel_frac'].items()}, 'whole', round(d['whole_iteration_frac'], 3), 'hbm6.3', round(d.get('hbm_frac_of_6.3TBps', 0), 3), 'fl/B', round(d.get('flop_per_byte', 0), 1)); print('   ', d['geometry'])
    else: print(l.rstrip())
"
done
