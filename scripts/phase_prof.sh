#!/bin/bash
# diagnostic: run bench with the -DSNMF_PROF build of the library (phase shares on stderr)
cp se_snmf_nat_amd/libsnmf_hip.so /tmp/libsnmf_hip.so.bak
cp scripts/prof_build/libsnmf_hip_prof.bin se_snmf_nat_amd/libsnmf_hip.so
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "SNMF_PROF|kernel_ms" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[SNMF'): last=l
    elif l.startswith('{'): d=json.loads(l); print(last.strip()); print({k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()})
"; }
run SNMF_X=1
run SNMF_PROF_W=1
run SNMF_PROF_W=1 SNMF_WSTATS_NL=0
cp /tmp/libsnmf_hip.so.bak se_snmf_nat_amd/libsnmf_hip.so
