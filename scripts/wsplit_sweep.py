"""Sweep of the row-group split of k_wstats (SNMF_WSTATS_X: modelled relative cost of the extra row per tile -> chunks of group 0 /
group 1): HIP-event time of the W statistics at C2, a11 and c4w per value.  Run on the GPU box."""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from se_snmf_nat_amd import Context, Plan

ctx = Context(0)
SH = {"C2": (257, 100000, 256, "full"), "a11": (513, 72000, 100, "full"), "c4w": (513, 100000, 100, "w")}
for name, (F, T, r, mode) in SH.items():
    rs = np.random.default_rng(1)
    V = (rs.gamma(0.5, 1.0, (F, 16)).astype(np.float32) @ rs.gamma(0.3, 1.0, (16, T)).astype(np.float32) + 1e-3)
    W0 = rs.random((F, r)); H0 = rs.random((r, T), dtype=np.float32)
    for x in (None, "0.02", "0.04", "0.06", "0.08", "0.10", "0.12", "0.15"):
        if x is None: os.environ.pop("SNMF_WSTATS_X", None)
        else: os.environ["SNMF_WSTATS_X"] = x
        kw = dict(h_update_ind=np.zeros(r, bool)) if mode == "w" else {}
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=600, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init()
        pl.run_async(150); ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timing(True); pl.run_async(40); ctx.sync()
            best = min(best, ctx.timing_get("wstats")[0]); ctx.timing(False)
        m = re.search(r"grid=\((\d+) chunks,(\d+) fgroups,\d+ kgroups; group-1 chunks (\d+)\)", pl.describe())
        print(f"{name} x={x}: chunks {m.group(1)}/{m.group(3)}  wstats {best*1e3:.1f} us", flush=True)
        pl.close()
