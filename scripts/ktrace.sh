#!/bin/bash
# quick per-kernel durations of bench.py under rocprofv3 --kernel-trace --stats (run on the GPU box). Usage: scripts/ktrace.sh <tag> [bench args]
TAG=${1:-kt}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$ST" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} min_us {float(r["MinNs"])/1e3:9.2f} max_us {float(r["MaxNs"])/1e3:9.2f}')
PY
cp $ST $GRAFT_REPO_ROOT/gpurun_out/kt_${TAG}_kernel_stats.csv
