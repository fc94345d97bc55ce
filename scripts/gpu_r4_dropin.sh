#!/bin/bash
# one gpurun call: the drop-in tests, then scripts/bench_dropin.py and bench.py (logs under gpurun_out/)
set -o pipefail
TAG=${1:-r4b}
SEL=${2:-"tests/test_dropin.py tests/test_frontend.py tests/test_abi.py"}
mkdir -p gpurun_out
echo "== pytest -m gpu $SEL"
timeout -k 10 900 python -X faulthandler -m pytest $SEL -m gpu -x -q --durations=8 -o faulthandler_timeout=300 > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
tail -15 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "== bench_dropin"
timeout -k 10 600 python scripts/bench_dropin.py ${3:-pcie a11 c2 c4 mel} > gpurun_out/${TAG}_dropin.jsonl 2> gpurun_out/${TAG}_dropin.err || { tail -20 gpurun_out/${TAG}_dropin.err; exit 1; }
python - <<PY
import json
for l in open("gpurun_out/${TAG}_dropin.jsonl"):
    d = json.loads(l); d.pop("note", None)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()})
PY
echo "== bench.py"
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "it/s", {k:round(v,4) for k,v in d["roofline"]["kernel_ms"].items()}, "frac", round(d["roofline"]["frac"],3), "dropin_ms", d.get("dropin_ms"), d.get("dropin"))
PY
