"""Round 6 development check of the small-rank family (csrc/snmf_smallr.h) on the GPU box: a few shapes against the fp64 oracle and
against the role pipelines (SNMF_HSTEP_SR=0 / SNMF_WSTATS_SR=0), then timings of both.   python scripts/sr_check.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.sparse_nmf_oracle import sparse_nmf as onmf
from se_snmf_nat_amd import Context, Plan, sparse_nmf


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


ctx = Context(0)
shapes = [(513, 20, 12000, "full"), (513, 30, 21000, "h"), (513, 10, 9000, "w"), (257, 32, 26000, "full"), (257, 64, 20000, "full"),
          (385, 33, 21157, "full"), (129, 50, 12000, "full"), (100, 20, 30000, "full"), (513, 1, 9000, "full"), (65, 8, 30011, "semi")]
bad = 0
for F, r, T, mode in shapes:
    rs = np.random.default_rng(F + r)
    V = rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3
    W0, H0 = rs.random((F, r)), rs.random((r, T))
    for sp in (5.0, rs.random(r) * 4, rs.random((r, T)) * 3):
        p = dict(cf="kl", sparsity=sp, max_iter=3, conv_eps=0, cost_check=1, init_w=W0, init_h=H0)
        if mode == "h":
            p["w_update_ind"] = np.zeros(r, bool)
        elif mode == "w":
            p["h_update_ind"] = np.zeros(r, bool)
        elif mode == "semi":
            p["w_update_ind"] = np.arange(r) >= r // 2
        w, h, o = sparse_nmf(V, p, ctx=ctx)
        wr, hr, orf = onmf(V, p)
        ec = float(np.max(np.abs(o["cost"] - orf["cost"]) / np.abs(orf["cost"])))
        ew, eh = rel(w, wr), rel(h, hr)
        ok = ew < 1e-4 and eh < 1e-4 and ec < 1e-5
        bad += not ok
        print(f"F={F} r={r} T={T} {mode} sp={'scalar' if np.isscalar(sp) else sp.shape}: relW {ew:.1e} relH {eh:.1e} cost {ec:.1e} {'ok' if ok else '<<< FAIL'}", flush=True)
        if len(sys.argv) > 1:
            break
print("failures:", bad)


def bench(F, T, r, iters=60, **kw):
    rs = np.random.default_rng(1)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=iters + 30, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
    pl.set_v(V); pl.set_w(rs.random((F, r))); pl.set_h(rs.random((r, T)).astype(np.float32)); pl.init()
    geo = pl.describe()
    pl.run_async(30); ctx.sync()
    ctx.timing(True)
    t = time.perf_counter(); pl.run_async(iters); ctx.sync(); dt = time.perf_counter() - t
    fam = {f: ctx.timing_get(f) for f in ("hstep", "wstats", "wfin", "reduce")}
    ctx.timing(False)
    pl.close()
    return iters / dt, {k: round(v[0] * 1e3, 1) for k, v in fam.items() if v[1]}, geo


for F, T, r, kw in [(513, 72000, 20, {}), (513, 72000, 30, dict(w_update_ind=np.zeros(30, bool))), (513, 72000, 10, dict(h_update_ind=np.zeros(10, bool))),
                    (257, 100000, 32, {}), (257, 100000, 64, {}), (513, 72000, 50, {})]:
    for sw in ("1", "0"):
        os.environ["SNMF_HSTEP_SR"] = sw
        os.environ["SNMF_WSTATS_SR"] = sw
        its, fam, geo = bench(F, T, r, **kw)
        print(f"F={F} T={T} r={r} {'H-only' if 'w_update_ind' in kw else 'W-only' if 'h_update_ind' in kw else 'full'} sr={sw}: {its:.0f} it/s  kernels us {fam}  | {'k_hstep_sr' in geo} {'k_wstats_sr' in geo}", flush=True)
