#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box via gpurun). Usage: scripts/prof.sh <tag>
set -x
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin > $OUT/bench_pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin > $OUT/bench_pmc2.json 2> $OUT/pmc2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin > $OUT/bench_pmc3.json 2> $OUT/pmc3.err
find $OUT -name "*.csv" | head -20
ls -la $OUT/*
