#!/bin/bash
# round 6: the stress build (-DSNMF_STRESS) on the lists that cover the objective fold and the shared tiles, then the new fuzz envelope twice more
mkdir -p gpurun_out
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 800 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_wfin.py tests/test_online.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r6n_stress.log 2>&1
echo "stress rc=$?" | tee -a gpurun_out/r6n_stress.log
tail -3 gpurun_out/r6n_stress.log
timeout -k 10 200 python scripts/fuzz_shapes.py 621 60 90 share > gpurun_out/r06_fuzz_share.log 2>&1; tail -2 gpurun_out/r06_fuzz_share.log
timeout -k 10 200 python scripts/fuzz_shapes.py 622 60 90 stop > gpurun_out/r06_fuzz_stop.log 2>&1; tail -2 gpurun_out/r06_fuzz_stop.log
