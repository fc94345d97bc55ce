#!/bin/bash
# round 6: does k_wfin pay for instruction fetches?  SQC instruction-cache counters per kernel on the Mel iteration (k_iter_sf, k_wfin alternate)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 -L > $OUT/counters.txt 2>&1; grep -i -o "SQC_[A-Z_]*ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_INST_CYCLES_VMEM\|SQC_INST[A-Z_]*" $OUT/counters.txt | sort -u | head -20
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 scripts/bench_f513.py mel --iters 20 > $OUT/b.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r6p/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
find $OUT -name "*_counter_collection.csv" -delete
