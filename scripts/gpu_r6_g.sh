#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r6k_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r6k_tests.log
tail -4 gpurun_out/r6k_tests.log
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 700 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_wfin.py tests/test_online.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r6k_stress.log 2>&1
echo "stress rc=$?" | tee -a gpurun_out/r6k_stress.log
tail -3 gpurun_out/r6k_stress.log
