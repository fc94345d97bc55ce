#!/bin/bash
# round 6, final sources: one more seed of every fuzz envelope (scripts/fuzz_shapes.py <seed> <cases> <seconds> <focus>)
mkdir -p gpurun_out
: > gpurun_out/r06_fuzz_campaign.log
for spec in "631 400 110 " "632 200 110 pipe" "633 200 90 r5" "634 60 110 big" "635 200 90 stop" "636 200 110 share"; do
  set -- $spec
  echo "== seed $1 focus '${4:-general}'" >> gpurun_out/r06_fuzz_campaign.log
  timeout -k 10 200 python scripts/fuzz_shapes.py $1 $2 $3 $4 > gpurun_out/fz.log 2>&1 || { tail -5 gpurun_out/fz.log; }
  grep "FAIL" gpurun_out/fz.log >> gpurun_out/r06_fuzz_campaign.log
  tail -2 gpurun_out/fz.log >> gpurun_out/r06_fuzz_campaign.log
done
cat gpurun_out/r06_fuzz_campaign.log
