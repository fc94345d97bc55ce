#!/bin/bash
# rocprofv3 passes for an arbitrary python script of this repo (run on the GPU box via gpurun):
#   scripts/prof_cmd.sh <tag> "<script + args of the trace pass>" "<script + args of the (shorter) counter passes>"
# kernel trace + stats, then three SEPARATE --pmc passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE: never combined with a trace).
# Output under gpurun_out/prof_<tag>/; scripts/summarize_prof.py <tag> "<command>" turns it into profiles/<tag>_*.
TAG=$1; TRACE_CMD=$2; PMC_CMD=${3:-$2}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "== trace: $TRACE_CMD"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $TRACE_CMD > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
echo "== pmc1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $PMC_CMD > $OUT/bench_pmc1.json 2> $OUT/pmc1.err || { tail -5 $OUT/pmc1.err; exit 1; }
echo "== pmc2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 $PMC_CMD > $OUT/bench_pmc2.json 2> $OUT/pmc2.err || { tail -5 $OUT/pmc2.err; exit 1; }
echo "== pmc3"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 $PMC_CMD > $OUT/bench_pmc3.json 2> $OUT/pmc3.err || { tail -5 $OUT/pmc3.err; exit 1; }
# the raw per-dispatch CSVs are large: keep the stats table and per-kernel averages only
python3 scripts/summarize_prof.py $TAG "$TRACE_CMD" > $OUT/summary.txt 2>&1 || { tail -5 $OUT/summary.txt; exit 1; }
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete
tail -30 $OUT/summary.txt
