run() { echo "== $*"; env "$@" python - <<'PY'
import os,sys,json,subprocess
sys.path.insert(0,'.')
from se_snmf_nat_amd import _lib, Context
import bench
ctx=Context(0); _lib.load().snmf_abi_version()
sys.argv=['bench.py','--steps','20','--warmup','3','--no-cpu-baseline']
import io,contextlib
buf=io.StringIO()
with contextlib.redirect_stdout(buf): bench.main()
d=json.loads(buf.getvalue().strip().splitlines()[-1]); print({k:round(v,4) for k,v in d['roofline']['kernel_ms'].items()})
PY
}
run SNMF_X=1
run SNMF_DBG_Q0=1
