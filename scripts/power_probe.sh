#!/bin/bash
# polls socket power / sclk while bench.py runs a long timed region (is the solver power-bound?)
TAG=${1:-pw}
mkdir -p gpurun_out
for rp in 0 1; do
  ( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/${TAG}_smi_rp$rp.log 2>&1 &
  SMI=$!
  SNMF_HSTEP_RP=$rp timeout -k 10 200 python bench.py --steps 12000 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_rp$rp.json 2> gpurun_out/${TAG}_bench_rp$rp.err
  wait $SMI
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_rp$rp.json").read().strip().splitlines()[-1])
print("rp=$rp", round(d["value"],1), "it/s", {k:round(v,4) for k,v in d["roofline"]["kernel_ms"].items()})
PY
  sort gpurun_out/${TAG}_smi_rp$rp.log | uniq -c | sort -rn | head -8
done
