#!/bin/bash
# round 6, final sources: the -m gpu suite on the product library, then the stress build on the pipelined lists
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/r06_gpu_tests.log
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 600 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_wfin.py tests/test_online.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r06_stress.log 2>&1
echo "stress rc=$?" | tee -a gpurun_out/r06_stress.log; tail -3 gpurun_out/r06_stress.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
