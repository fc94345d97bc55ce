#!/usr/bin/env python3
"""k_iter_sf (H step + W statistics in one launch) against the two launches: bits, then time.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from se_snmf_nat_amd import Context, Plan

def run(ctx, V, W0, H0, r, iters, env, sparsity=1.0, conv_eps=0.0):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        F, T = V.shape
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=iters, conv_eps=conv_eps, cost_check=True, sparsity=sparsity)
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
        out = (pl.get_h(np.float32), pl.get_w(), pl.describe(), pl.get_objective())
        pl.close()
        return out
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

ctx = Context(0)
bad = 0
for F, r, T, sp in [(64, 100, 20000, 1.0), (64, 128, 9000, 5.0), (40, 70, 33000, 1.0), (64, 96, 8231, np.linspace(0.5, 2.0, 96)), (64, 100, 72000, 5.0)]:
    rs = np.random.default_rng(F * 1000 + r)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r)); H0 = rs.random((r, T)).astype(np.float32)
    a = run(ctx, V, W0, H0, r, 4, {"SNMF_ITER_SF": "1"}, sparsity=sp)
    b = run(ctx, V, W0, H0, r, 4, {"SNMF_ITER_SF": "0"}, sparsity=sp)
    assert "k_iter_sf" in a[2], a[2]
    assert "k_iter_sf" not in b[2] and "k_hstep_sf" in b[2], b[2]
    eh, ew = np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1])
    dc = max(abs(x - y) / abs(y) for x, y in zip(a[3][1], b[3][1]) if y != 0)
    ok = eh and ew and dc < 1e-9 and a[3][2] == b[3][2]
    bad += not ok
    print(f"F={F} r={r} T={T}: H bit-equal {eh} W bit-equal {ew} rel cost diff {dc:.2e} n_it {a[3][2]}/{b[3][2]} {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
