#!/bin/bash
# round 6: the H-only loop's objective fold on the H step (SNMF_HFOLD): tests, then A/B on the H-only shapes + the headline (unchanged path)
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_wfin.py tests/test_gpu_pipelined_vs_plain.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r6j_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6j_tests.log
timeout -k 10 300 python scripts/bench_f513.py c2 a11 c4h melh tw30h mel > gpurun_out/r6j_fold.jsonl 2> gpurun_out/r6j_fold.err; echo "fold rc=$?"
SNMF_HFOLD=0 timeout -k 10 300 python scripts/bench_f513.py c4h melh tw30h > gpurun_out/r6j_nofold.jsonl 2> gpurun_out/r6j_nofold.err; echo "nofold rc=$?"
python - <<'PY'
import json
for f in ("r6j_fold", "r6j_nofold"):
    print(f)
    for l in open("gpurun_out/%s.jsonl" % f):
        x = json.loads(l); print(" ", x["shape"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()})
PY
