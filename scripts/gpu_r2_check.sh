#!/bin/bash
# one gpurun call: GPU test suite, then the bench with the role-pipeline H step off / on (logs under gpurun_out/)
set -o pipefail
TAG=${1:-r2a}
mkdir -p gpurun_out
echo "== pytest -m gpu" 
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q --durations=15 -o faulthandler_timeout=150 > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
for rp in 0 1; do
  echo "== bench SNMF_HSTEP_RP=$rp"
  SNMF_HSTEP_RP=$rp timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_rp$rp.json 2> gpurun_out/${TAG}_bench_rp$rp.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_rp$rp.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "it/s", {k:round(v,4) for k,v in d["roofline"]["kernel_ms"].items()}, "frac", round(d["roofline"]["frac"],3), d.get("cost_vs_oracle"))
PY
done
