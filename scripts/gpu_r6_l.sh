#!/bin/bash
# round 6: the shared last tile of k_hstep_sf: tests, then melh A/B (SNMF_HSTEP_SPLIT=0 = every tile whole)
mkdir -p gpurun_out
timeout -k 10 200 python scripts/bench_f513.py melh > gpurun_out/r6l_share.jsonl 2> gpurun_out/r6l_share.err; echo "share rc=$?"
SNMF_HSTEP_SPLIT=0 timeout -k 10 200 python scripts/bench_f513.py melh > gpurun_out/r6l_whole.jsonl 2> gpurun_out/r6l_whole.err; echo "whole rc=$?"
python - <<'PY'
import json
for f in ("r6l_share", "r6l_whole"):
    for l in open("gpurun_out/%s.jsonl" % f):
        x = json.loads(l); print(f, x["shape"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()}, x["geometry"][60:230])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_parity.py tests/test_gpu_wfin.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize_shapes.py -m gpu -q > gpurun_out/r6l_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r6l_tests.log
