#!/bin/bash
# Which kernel instantiations do the reference's settings launch?  One rocprofv3 --kernel-trace --stats invocation per shape of
# scripts/reach_one.py (run on the GPU box) -> gpurun_out/reach/<shape>_kernels.txt -> profiles/r06_reachable.json (scripts/gen_resources.py reads it).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/reach
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for s in $(python3 -c "
import re
src = open('scripts/reach_one.py').read()
print(' '.join(re.findall(r'\"(\w+)\": \(', src)))"); do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$s -- python3 scripts/reach_one.py $s > $OUT/$s.log 2> $OUT/$s.err || { echo "FAILED $s"; tail -3 $OUT/$s.err; continue; }
  f=$(find $OUT/$s -name "*_kernel_stats.csv" | head -1)
  python3 - "$f" "$s" >> $OUT/all.jsonl <<PY
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = sorted({r["Name"].replace("void snmf::", "").replace("snmf::", "").split("(")[0] for r in rows if "snmf" in r["Name"] or r["Name"].startswith("void k_") or "k_" in r["Name"]})
print(json.dumps({"shape": sys.argv[2], "kernels": names}))
PY
  rm -rf $OUT/$s
  echo "== $s done"
done
python3 - <<PY
import json
out = {}
for l in open("$OUT/all.jsonl"):
    d = json.loads(l)
    for k in d["kernels"]:
        out.setdefault(k, []).append(d["shape"])
json.dump(out, open("$OUT/reachable.json", "w"), indent=1, sort_keys=True)
print(len(out), "kernel instantiations reached")
PY
