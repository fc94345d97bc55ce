#!/bin/bash
# round 6: how much of the tile-per-wave (small-F) family's time is round quantisation?  The same shapes at frame counts that fill every SIMD evenly.
mkdir -p gpurun_out
: > gpurun_out/r6k_quant.jsonl
for T in 98304 100000 65536 72000; do
  SNMF_BENCH_T=$T timeout -k 10 120 python scripts/bench_f513.py mel melh melw >> gpurun_out/r6k_quant.jsonl 2>/dev/null || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r6k_quant.jsonl"):
    x = json.loads(l); print(x["shape"], x["T"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()}, "us/it", round(x["ms_per_iteration"]*1e3,1), "ns per 32-frame tile", round(x["ms_per_iteration"]*1e6/(x["T"]/32),2))
PY
