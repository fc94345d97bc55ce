#!/bin/bash
# one gpurun call at the end of a round: the whole -m gpu suite, then the committed bench lines
#   scripts/gpu_round_final.sh <tag>  ->  gpurun_out/<tag>_{pytest.log,bench.json,bench_f513.jsonl,dropin.jsonl,smoke.log}
set -o pipefail
TAG=${1:-r05}
mkdir -p gpurun_out
if [ "${2:-tests}" != "notests" ]; then
timeout -k 10 1000 python -X faulthandler -m pytest tests -m gpu -x -q --durations=8 -o faulthandler_timeout=400 > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?; tail -6 gpurun_out/${TAG}_pytest.log; [ $rc -ne 0 ] && exit $rc
fi
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_smoke.log; exit 1; }
echo "== bench.py"
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
tail -c 1500 gpurun_out/${TAG}_bench.json
echo "== bench_f513"
timeout -k 10 400 python scripts/bench_f513.py a11 c4h c4w c5 mel melh melw smallr tw20 tw30h im50 > gpurun_out/${TAG}_bench_f513.jsonl 2> gpurun_out/${TAG}_bench_f513.err || { tail -20 gpurun_out/${TAG}_bench_f513.err; exit 1; }
cut -c 1-420 gpurun_out/${TAG}_bench_f513.jsonl
echo "== bench_dropin"
timeout -k 10 900 python scripts/bench_dropin.py pcie a11 c2 c4 mel c4mel c4m > gpurun_out/${TAG}_dropin.jsonl 2> gpurun_out/${TAG}_dropin.err || { tail -20 gpurun_out/${TAG}_dropin.err; exit 1; }
cut -c 1-500 gpurun_out/${TAG}_dropin.jsonl
echo "== bench_online"
timeout -k 10 300 python scripts/bench_online.py > gpurun_out/${TAG}_bench_online.jsonl 2> gpurun_out/${TAG}_bench_online.err || { tail -20 gpurun_out/${TAG}_bench_online.err; exit 1; }
cut -c 1-400 gpurun_out/${TAG}_bench_online.jsonl
