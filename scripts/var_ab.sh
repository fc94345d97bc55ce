#!/bin/bash
# A/B of experiment builds (scripts/build_snap.sh var:NAME:FLAGS) on the bench, one process each, in the given order
# usage (on the GPU box): scripts/var_ab.sh NAME[@VAR=VALUE]...      (NAME "prod" = the product library)
cd "$(dirname "$0")/.."
for spec in "$@"; do
  n=${spec%%@*}; ev=""; [ "$spec" != "$n" ] && ev=${spec#*@}   # NAME or NAME@VAR=VALUE (one environment setting)
  [ -n "$ev" ] && export "$ev"
  if [ "$n" = prod ]; then unset SNMF_LIB_PATH; else export SNMF_LIB_PATH="$PWD/scripts/prof_build/libsnmf_$n.so"; fi
  timeout -k 10 200 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline > /tmp/ab.out 2> /tmp/ab.err || { echo "$spec: FAILED"; tail -3 /tmp/ab.err; continue; }
  python - "$spec" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.out').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:>10}: {d['value']:.1f} it/s", {k: round(v, 4) for k, v in d['roofline']['kernel_ms'].items()}, "cost_vs_oracle", d.get("cost_vs_oracle"), flush=True)
PY
  [ -n "$ev" ] && unset "${ev%%=*}"
done
