#!/bin/bash
# dynamic instruction mix of the kernels of an arbitrary script of this repo (rocprofv3 PMC, three passes, no trace):
#   scripts/pmc_insts_cmd.sh <tag> "<script + args>"      -> gpurun_out/pmci_<tag>/summary.txt
TAG=$1; CMD=$2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmci_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $CMD > $OUT/b1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/p2 -- python3 $CMD > $OUT/b2.json 2> $OUT/p2.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 $CMD > $OUT/b3.json 2> $OUT/p3.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p4 -- python3 $CMD > $OUT/b4.json 2> $OUT/p4.err
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections
for p in ("p1","p2","p3","p4"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs:
        print(p, "no csv; err tail:", open("$OUT/%s.err" % p).read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k in acc:
        if len(n[k]) < 3: continue
        print(p, k, len(n[k]), {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
find $OUT -name "*_counter_collection.csv" -delete
cat $OUT/summary.txt
