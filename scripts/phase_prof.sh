#!/bin/bash
# diagnostic: run bench with the -DSNMF_PROF build of the library (phase shares on stderr).
# The diagnostic build lives at its OWN path (scripts/prof_build/libsnmf_hip_prof.so, git-ignored, built here when
# missing) and is selected with SNMF_LIB_PATH; the product library se_snmf_nat_amd/libsnmf_hip.so is never touched.
set -e
cd "$(dirname "$0")/.."
PROF=scripts/prof_build/libsnmf_hip_prof.so
python scripts/build_variant.py hip_prof -DSNMF_PROF > /dev/null   # (rebuilds only the translation units whose sources changed)
export SNMF_LIB_PATH="$PWD/$PROF"
run() { echo "== $*"; env "$@" python bench.py --steps 200 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "SNMF_PROF|kernel_ms" | python -c "
import sys,json
last=''
for l in sys.stdin:
    if l.startswith('[SNMF'): last=l
    elif l.startswith('{'): d=json.loads(l); print(last.strip()); print({k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()})
"; }
if [ $# -gt 0 ]; then for v in "$@"; do run $v; done; exit 0; fi
run SNMF_HSTEP_RP=1
run SNMF_HSTEP_RP=0
run SNMF_PROF_W=1
