#!/bin/bash
# round 6: the shared remainder tile of k_iter_sf: A/B on the Mel training shape, then the tests that cover it
mkdir -p gpurun_out
timeout -k 10 200 python scripts/bench_f513.py mel mel288 > gpurun_out/r6u_share.jsonl 2> gpurun_out/r6u_share.err; echo "share rc=$?"
SNMF_HSTEP_SPLIT=0 timeout -k 10 200 python scripts/bench_f513.py mel mel288 > gpurun_out/r6u_whole.jsonl 2> gpurun_out/r6u_whole.err; echo "whole rc=$?"
python - <<'PY'
import json
for f in ("r6u_share", "r6u_whole"):
    for l in open("gpurun_out/%s.jsonl" % f):
        x = json.loads(l); print(f, x["shape"], x["T"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()})
PY
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_pipelined_vs_plain.py -m gpu -q -x > gpurun_out/r6u_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r6u_tests.log
