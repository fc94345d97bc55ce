#!/bin/bash
# round 6: k_wfin / k_wapply load w_ind[k] and lambda_k[k] at the top of the kernel (two dependent round trips off the critical path): tests + times
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_wfin.py tests/test_gpu_parity.py tests/test_gpu_multi_abi.py -m gpu -q -x > gpurun_out/r6r_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6r_tests.log
timeout -k 10 300 python scripts/bench_f513.py c2 a11 mel melw tw20 smallr c4w > gpurun_out/r6r.jsonl 2> gpurun_out/r6r.err; echo "rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r6r.jsonl"):
    x = json.loads(l); print(x["shape"], x["T"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()})
PY
