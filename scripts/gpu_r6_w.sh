#!/bin/bash
# round 6, final sources: a second extra seed of every fuzz envelope (after the shared tile of k_iter_sf) (scripts/fuzz_shapes.py <seed> <cases> <seconds> <focus>)
mkdir -p gpurun_out
: > gpurun_out/r06_fuzz_campaign2.log
for spec in "651 400 110 " "652 200 110 pipe" "653 200 90 r5" "654 60 110 big" "655 200 90 stop" "656 200 110 share"; do
  set -- $spec
  echo "== seed $1 focus '${4:-general}'" >> gpurun_out/r06_fuzz_campaign2.log
  timeout -k 10 200 python scripts/fuzz_shapes.py $1 $2 $3 $4 > gpurun_out/fz.log 2>&1 || { tail -5 gpurun_out/fz.log; }
  grep "FAIL" gpurun_out/fz.log >> gpurun_out/r06_fuzz_campaign2.log
  tail -2 gpurun_out/fz.log >> gpurun_out/r06_fuzz_campaign2.log
done
cat gpurun_out/r06_fuzz_campaign2.log
