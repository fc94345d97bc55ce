#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scripts/sr_check.py quick > gpurun_out/r6d_sr.log 2>&1; echo "rc=$?" >> gpurun_out/r6d_sr.log
tail -24 gpurun_out/r6d_sr.log
for st in 0 600 2600; do echo "SNMF_SR_STAG=$st"; SNMF_SR_STAG=$st timeout -k 10 200 python scripts/sr_check.py quick 2>&1 | grep "sr=1"; done | tee gpurun_out/r6d_stag.log
