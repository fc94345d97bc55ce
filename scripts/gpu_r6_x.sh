#!/bin/bash
# round 6: k_wfin's latency chain -- DPP steps in the epilogue's column sums, the objective's tree by shuffles: stamps, tests, kernel times
mkdir -p gpurun_out
SNMF_LIB_PATH=scripts/prof_build/libsnmf_wfprof.so timeout -k 10 200 python scripts/wfin_prof.py 2>&1 | grep -v "SNMF_PROF\|amdgpu.ids" > gpurun_out/r06_wfin_stamps2.log; cat gpurun_out/r06_wfin_stamps2.log
timeout -k 10 500 python -m pytest tests/test_gpu_wfin.py tests/test_gpu_parity.py tests/test_gpu_multi_abi.py -m gpu -q -x > gpurun_out/r6x_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6x_tests.log
timeout -k 10 300 python scripts/bench_f513.py c2 a11 mel melw tw20 smallr > gpurun_out/r6x.jsonl 2> gpurun_out/r6x.err; echo "rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r6x.jsonl"):
    x = json.loads(l); print(x["shape"], x["T"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()})
PY
