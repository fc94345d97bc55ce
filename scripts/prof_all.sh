#!/bin/bash
# Every committed rocprofv3 summary of a round in one gpurun call (run on the GPU box):
#   scripts/prof_all.sh <tag> <source commit>   ->  gpurun_out/profiles_<tag>*/ (copy into profiles/)
# C2 through bench.py (scripts/prof.sh), then the four F = 513 shapes through scripts/bench_f513.py (scripts/prof_cmd.sh).
TAG=${1:-r03}; export SNMF_SOURCE_COMMIT=${2:-unknown}
bash scripts/prof.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 || exit 1
python3 scripts/summarize_prof.py $TAG > gpurun_out/prof_${TAG}_summary.txt 2>&1 || { tail -5 gpurun_out/prof_${TAG}_summary.txt; exit 1; }
rm -rf gpurun_out/profiles_$TAG; mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.csv profiles/${TAG}_traffic.json profiles/${TAG}_mfma_util.txt profiles/${TAG}_bench_trace.json gpurun_out/profiles_$TAG/
find gpurun_out/prof_$TAG -name "*_counter_collection.csv" -delete; find gpurun_out/prof_$TAG -name "*_kernel_trace.csv" -delete
echo "== C2 done"; head -6 profiles/${TAG}_kernel_stats.csv
for s in a11 c4h c4w c5 mel melh melw smallr tw20; do
  it=40; [ $s = c5 ] && it=6
  bash scripts/prof_cmd.sh ${TAG}_f513_$s "scripts/bench_f513.py $s" "scripts/bench_f513.py $s --iters $it" > gpurun_out/prof_${TAG}_f513_$s.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}_f513_$s.log; exit 1; }
  echo "== $s done"; head -5 profiles/${TAG}_f513_${s}_kernel_stats.csv
done
