#!/bin/bash
# round 6, GPU pass A: the -m gpu suite on the product library, then the hand-off stress build (-DSNMF_STRESS) on the pipelined lists
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r6a_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r6a_tests.log
tail -3 gpurun_out/r6a_tests.log
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 500 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_wfin.py "tests/test_online.py" -m gpu -q > gpurun_out/r6a_stress.log 2>&1
echo "stress rc=$?" | tee -a gpurun_out/r6a_stress.log
tail -3 gpurun_out/r6a_stress.log
