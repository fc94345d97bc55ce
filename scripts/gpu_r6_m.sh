#!/bin/bash
# round 6: shared tiles of the small-F family (k_hstep_sf, k_wstats_sf): A/B on the Mel H-only / W-only shapes, then the GPU suite
mkdir -p gpurun_out
timeout -k 10 200 python scripts/bench_f513.py melh melw mel > gpurun_out/r6m_share.jsonl 2> gpurun_out/r6m_share.err; echo "share rc=$?"
SNMF_HSTEP_SPLIT=0 timeout -k 10 200 python scripts/bench_f513.py melh melw > gpurun_out/r6m_whole.jsonl 2> gpurun_out/r6m_whole.err; echo "whole rc=$?"
SNMF_BENCH_T=72000 timeout -k 10 200 python scripts/bench_f513.py melh melw >> gpurun_out/r6m_share.jsonl 2>> gpurun_out/r6m_share.err
SNMF_BENCH_T=72000 SNMF_HSTEP_SPLIT=0 timeout -k 10 200 python scripts/bench_f513.py melh melw >> gpurun_out/r6m_whole.jsonl 2>> gpurun_out/r6m_whole.err
python - <<'PY'
import json
for f in ("r6m_share", "r6m_whole"):
    for l in open("gpurun_out/%s.jsonl" % f):
        x = json.loads(l); print(f, x["shape"], x["T"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()})
PY
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r6m_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r6m_tests.log
