#!/bin/bash
# one gpurun call: selected GPU tests (or all), then the bench with the split last round on / off (logs under gpurun_out/)
set -o pipefail
TAG=${1:-r3a}
SEL=${2:-tests}
mkdir -p gpurun_out
echo "== pytest -m gpu $SEL"
timeout -k 10 1000 python -X faulthandler -m pytest $SEL -m gpu -x -q --durations=10 -o faulthandler_timeout=200 > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
tail -8 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
for sp in 1 0; do
  echo "== bench SNMF_HSTEP_SPLIT=$sp"
  SNMF_HSTEP_SPLIT=$sp timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_sp$sp.json 2> gpurun_out/${TAG}_bench_sp$sp.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_sp$sp.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "it/s", {k:round(v,4) for k,v in d["roofline"]["kernel_ms"].items()}, "frac", round(d["roofline"]["frac"],3), d.get("cost_vs_oracle"))
print(d["config"]["geometry"])
PY
done
