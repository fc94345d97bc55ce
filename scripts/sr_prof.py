"""Phase stamps of k_hstep_sr (diagnostic -DSNMF_PROF build): SNMF_LIB_PATH=scripts/prof_build/libsnmf_hip_prof.so python scripts/sr_prof.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from se_snmf_nat_amd import Context, Plan

F, T, r = 513, 72000, 20
ctx = Context(0)
rs = np.random.default_rng(1)
V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
os.environ["SNMF_PROF_DUMP"] = "/tmp/sr_prof.bin"
pl = Plan(ctx, F, T, r, beta=1.0, max_iter=40, conv_eps=0.0, cost_check=True, sparsity=5.0, w_update_ind=np.zeros(r, bool))
pl.set_v(V); pl.set_w(rs.random((F, r))); pl.set_h(rs.random((r, T)).astype(np.float32)); pl.init()
pl.run_async(30); ctx.sync()
pl.close()
x = np.fromfile("/tmp/sr_prof.bin", dtype=np.uint64).reshape(-1, 12).astype(np.float64)
x = x[: 256 * 8]
names = ["load issue", "P1", "ratio", "P2", "extra row", "wait buffer free", "write partial", "wait partials", "finish slice", "loop top/copies", "-", "-"]
tot = x.sum(1)
print("cycles per wave: mean %.0f  min %.0f  max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    print(f"  {n:18s} {x[:, i].mean():9.0f}  {100 * x[:, i].sum() / x.sum():5.1f} %   per wave-of-workgroup:", " ".join("%.0f" % x[w::8, i].mean() for w in range(8)))
