#!/bin/bash
TAG=r06; export SNMF_SOURCE_COMMIT=$1
bash scripts/prof.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 || exit 1
python3 scripts/summarize_prof.py $TAG > gpurun_out/prof_${TAG}_summary.txt 2>&1 || { tail -5 gpurun_out/prof_${TAG}_summary.txt; exit 1; }
rm -rf gpurun_out/profiles_$TAG; mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.csv profiles/${TAG}_traffic.json profiles/${TAG}_mfma_util.txt profiles/${TAG}_bench_trace.json gpurun_out/profiles_$TAG/
find gpurun_out/prof_$TAG -name "*_counter_collection.csv" -delete; find gpurun_out/prof_$TAG -name "*_kernel_trace.csv" -delete
head -6 profiles/${TAG}_kernel_stats.csv
bash scripts/reach.sh > gpurun_out/reach.log 2>&1; tail -3 gpurun_out/reach.log
