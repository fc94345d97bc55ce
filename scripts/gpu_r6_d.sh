#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r6g_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r6g_tests.log
tail -12 gpurun_out/r6g_tests.log
for sw in 1 0; do
  SNMF_HSTEP_SR=$sw SNMF_WSTATS_SR=$sw timeout -k 10 200 python scripts/bench_f513.py tw20 tw30h smallr 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('sr=$sw', d['shape'], round(d['iterations_per_s']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})
"
done | tee gpurun_out/r6g_smallr.log
