#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python scripts/sr_check.py > gpurun_out/r6b_sr.log 2>&1; echo "sr rc=$?" >> gpurun_out/r6b_sr.log
tail -25 gpurun_out/r6b_sr.log
timeout -k 10 200 python scripts/fuzz_case_probe.py 605 "" 9 > gpurun_out/r6b_probe.log 2>&1; tail -8 gpurun_out/r6b_probe.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r6b_bench.json 2> gpurun_out/r6b_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r6b_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('from_random_start'), d.get('from_random_start_detail'))"
