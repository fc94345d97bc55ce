#!/bin/bash
# round 6: k_iter_sf's shared remainder tile under the stress build + more fuzz on the envelopes that reach it
mkdir -p gpurun_out
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 800 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_wfin.py tests/test_online.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r06_stress.log 2>&1
echo "stress rc=$?" | tee -a gpurun_out/r06_stress.log; tail -3 gpurun_out/r06_stress.log
timeout -k 10 200 python scripts/fuzz_shapes.py 641 100 120 share > gpurun_out/r06_fuzz_share.log 2>&1; tail -2 gpurun_out/r06_fuzz_share.log
timeout -k 10 200 python scripts/fuzz_shapes.py 642 100 100 r5 > gpurun_out/r06_fuzz_r5.log 2>&1; tail -2 gpurun_out/r06_fuzz_r5.log
SNMF_LIB_PATH=scripts/prof_build/libsnmf_stress.so timeout -k 10 200 python scripts/fuzz_shapes.py 643 100 100 share > gpurun_out/r06_fuzz_share_stress.log 2>&1; tail -2 gpurun_out/r06_fuzz_share_stress.log
