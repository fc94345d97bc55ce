#!/bin/bash
# round 6, final measurement pass: the bench line, every full-size shape, the drop-in calls, the online loop, two more fuzz seeds
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"
timeout -k 10 400 python scripts/bench_f513.py c2 a11 c4h c4w c5 mel mel288 melh melw smallr tw20 tw30h tw10w im50 > gpurun_out/r06_bench_f513.jsonl 2> gpurun_out/r06_bench_f513.err; echo "f513 rc=$?"
timeout -k 10 300 python scripts/bench_f513.py c2 a11 mel tw20 smallr --no-cost > gpurun_out/r06_bench_f513_nocost.jsonl 2>/dev/null; echo "nocost rc=$?"
timeout -k 10 400 python scripts/bench_dropin.py pcie a11 c2 c4 mel c4mel c4m > gpurun_out/r06_dropin.jsonl 2> gpurun_out/r06_dropin.err; echo "dropin rc=$?"
timeout -k 10 200 python scripts/bench_online.py --seconds 12 > gpurun_out/r06_bench_online.jsonl 2> gpurun_out/r06_online.err; echo "online rc=$?"
timeout -k 10 300 python scripts/fuzz_shapes.py 611 200 120 pipe > gpurun_out/r06_fuzz_pipe.log 2>&1; tail -2 gpurun_out/r06_fuzz_pipe.log
timeout -k 10 300 python scripts/fuzz_shapes.py 612 300 120 > gpurun_out/r06_fuzz.log 2>&1; tail -2 gpurun_out/r06_fuzz.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["from_random_start"], d["roofline"]["traffic_source"]["traffic_stale"], d.get("dropin_ms"), d["cpu_baseline"]["value"])
for l in open("gpurun_out/r06_bench_f513.jsonl"):
    x = json.loads(l); print(x["shape"], round(x["iterations_per_s"]), {k: round(v * 1e3, 1) for k, v in x["kernel_ms"].items()}, {k: round(v, 3) for k, v in x["kernel_frac"].items()})
PY
