#!/usr/bin/env python3
"""k_hstep_m (merged roles) against k_hstep_rp and the plain k_hstep: bits, then time.  Run on the GPU box:
    python scripts/hm_ab.py [--time-only]
SNMF_HSTEP_M / SNMF_HSTEP_RP / SNMF_HSTEP_SPLIT are read when a plan is created, so all variants run in this one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from se_snmf_nat_amd import Context, Plan  # noqa: E402


def run(ctx, V, W0, H0, r, *, h_only, iters, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        F, T = V.shape
        kw = dict(w_update_ind=np.zeros(r, bool)) if h_only else {}
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=iters, conv_eps=0.0, cost_check=True, sparsity=1.0, **kw)
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
        out = (pl.get_h(np.float32), pl.get_w(), pl.describe(), pl.get_objective()[1])
        pl.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def bits(ctx):
    bad = 0
    for F, r, T in [(257, 256, 30000), (257, 256, 9000), (256, 230, 17000), (256, 256, 8231), (257, 225, 40000 + 7), (257, 256, 8192 + 32 * 256)]:
        rs = np.random.default_rng(F * 1000 + r)
        V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
        W0 = rs.random((F, r))
        H0 = rs.random((r, T)).astype(np.float32)
        for h_only, iters in ((True, 2), (False, 3)):
            hm = run(ctx, V, W0, H0, r, h_only=h_only, iters=iters, env={"SNMF_HSTEP_M": "1"})
            pl = run(ctx, V, W0, H0, r, h_only=h_only, iters=iters, env={"SNMF_HSTEP_M": "0", "SNMF_HSTEP_RP": "0"})
            assert "k_hstep_m" in hm[2], hm[2]
            assert "k_hstep_m" not in pl[2] and "k_hstep_rp" not in pl[2]
            eq_h = np.array_equal(hm[0], pl[0])
            dw = np.abs(hm[1] - pl[1]).max() / np.abs(pl[1]).max()
            dobj = max(abs(a - b) / abs(b) for a, b in zip(hm[3], pl[3]) if b != 0)
            ok = (eq_h if h_only else True) and dw <= 2e-6 and dobj <= 1e-6
            bad += not ok
            print(f"F={F} r={r} T={T} h_only={h_only}: H bit-equal {eq_h}  max|dH| {np.abs(hm[0] - pl[0]).max():.3e}  relW {dw:.2e}  relobj {dobj:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    return bad


def timeit(ctx, env, F=257, T=100_000, r=256, settle=150, K=40, full=True):
    import bench
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        V, W0, H0 = bench.make_problem(F, T, r)
        kw = {} if full else dict(w_update_ind=np.zeros(r, bool))
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=settle + K + 1, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
        pl.set_v(V.astype(np.float32)); pl.set_w(W0); pl.set_h(H0.astype(np.float32)); pl.init()
        pl.run_async(settle); ctx.sync()
        ctx.timing(True)
        t = time.perf_counter()
        pl.run_async(K); ctx.sync()
        dt = time.perf_counter() - t
        fam = {f: ctx.timing_get(f) for f in ("hstep", "wstats", "wfin")}
        ctx.timing(False)
        cost = pl.get_objective()[1]
        d = pl.describe()
        pl.close()
        return dt / K * 1e3, fam, [c for c in cost if c != 0][-1], d
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == "__main__":
    ctx = Context(0)
    bad = 0
    if "--time-only" not in sys.argv:
        bad = bits(ctx)
    for rep in range(2):
        for name, env in (("hm", {"SNMF_HSTEP_M": "1"}), ("rp split", {"SNMF_HSTEP_M": "0"}), ("rp unsplit", {"SNMF_HSTEP_M": "0", "SNMF_HSTEP_SPLIT": "0"})):
            ms, fam, cost, d = timeit(ctx, env)
            print(f"C2 {name:10s}: {ms:.4f} ms/it  hstep {fam['hstep'][0]:.4f} ms = {4 * 257 * 1e5 * 256 / fam['hstep'][0] / 1e9 / 157.3:.3f} of peak   wstats {fam['wstats'][0]:.4f}  wfin {fam['wfin'][0]:.4f}  cost {cost:.8e}", flush=True)
    sys.exit(1 if bad else 0)
