#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()}, round(d['roofline']['frac'],3))"; }
run SNMF_X=1
run SNMF_HSTEP_CFG=8x1x8
run SNMF_HSTEP_CFG=8x1x4
run SNMF_HSTEP_CFG=8x1x8
