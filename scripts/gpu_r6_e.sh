#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r6h_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r6h_tests.log
tail -5 gpurun_out/r6h_tests.log
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 600 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_wfin.py tests/test_online.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r6h_stress.log 2>&1
echo "stress rc=$?" | tee -a gpurun_out/r6h_stress.log
tail -4 gpurun_out/r6h_stress.log
# round 5's race put back (SNMF_HSTEP_SR=0: r <= 32 on the role pipeline again, where the race lived): the fuzz test must FAIL
SNMF_HSTEP_SR=0 SNMF_LIB_PATH=scripts/prof_build/libsnmf_r5race.so timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r6h_r5race.log 2>&1
echo "r5race rc=$? (expected: 1)" | tee -a gpurun_out/r6h_r5race.log
grep -E "FAIL|passed|failed" gpurun_out/r6h_r5race.log | cut -c1-220 | tail -8
