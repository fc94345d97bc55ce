#!/bin/bash
# rocprofv3 kernel trace of the online loop (run on the GPU box via gpurun). Usage: scripts/prof_online.sh <tag>
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_online_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_online.py --seconds 6 > $OUT/bench_online_trace.json 2> $OUT/trace.err
find $OUT -name "*kernel_stats.csv" | head -3
