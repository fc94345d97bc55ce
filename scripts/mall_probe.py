"""How much of the headline H step's time is the memory side?  k_hstep_rp at C2's shape inside the full iteration (V and H re-read
after ~340 MB of other traffic) against the same kernel in an H-only loop (V re-read after ~200 MB: it stays in the 256 MB cache behind
the L2s) and on a quarter of the frames (everything stays).  python scripts/mall_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from se_snmf_nat_amd import Context, Plan

ctx = Context(0)
F, r = 257, 256
for T, mode in ((100000, "full"), (100000, "h"), (25000, "full"), (25000, "h")):
    rs = np.random.default_rng(0)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    kw = dict(w_update_ind=np.zeros(r, bool)) if mode == "h" else {}
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=1000, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
    pl.set_v(V); pl.set_w(rs.random((F, r))); pl.set_h(rs.random((r, T)).astype(np.float32)); pl.init()
    pl.run_async(150); ctx.sync()
    ctx.timing(True); pl.run_async(100); ctx.sync()
    fam = {f: ctx.timing_get(f) for f in ("hstep", "wstats", "wfin")}
    ctx.timing(False)
    t = time.perf_counter(); pl.run_async(100); ctx.sync(); dt = time.perf_counter() - t
    print(f"T={T} {mode}: {100 / dt:.0f} it/s; per launch us:", {k: round(v[0] * 1e3, 1) for k, v in fam.items() if v[1]}, "| hstep per 32-frame tile ns:", round(fam["hstep"][0] * 1e6 / (T / 32), 1))
    pl.close()
