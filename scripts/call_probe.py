#!/usr/bin/env python3
"""Where a drop-in call's time goes beyond transfers and the steady-state solve: the steps of snmf_sparse_nmf_oop_* one by one
through the plan API (C2, 20 iterations, host fp64 arrays), each followed by a stream sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from se_snmf_nat_amd import Context, Plan, sparse_nmf

ctx = Context(0)
F, T, r, K = 257, 100_000, 256, 20
V, W0, H0 = bench.make_problem(F, T, r)
Vh, Wh, Hh = (np.asfortranarray(M, dtype=np.float64) for M in (V, W0, H0))
pd = dict(cf="kl", sparsity=5.0, max_iter=K, conv_eps=0, cost_check=1, init_w=Wh, init_h=Hh)
sparse_nmf(Vh[:, :4096], dict(pd, init_h=Hh[:, :4096], max_iter=2), ctx=ctx)
for rep in range(3):
    t = [time.perf_counter()]
    def lap(): ctx.sync(); t.append(time.perf_counter())
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=K, conv_eps=0.0, cost_check=True, sparsity=5.0); lap()
    pl.set_v(Vh); lap()
    pl.set_w(Wh); lap()
    pl.set_h(Hh); lap()
    pl.init(); lap()
    pl.run(); lap()
    w = pl.get_w(); lap()
    ctx.xfer_stats(reset=True)
    h = pl.get_h(); lap()
    st = ctx.xfer_stats()
    t1 = time.perf_counter(); h2 = pl.get_h(); dt2 = time.perf_counter() - t1
    st2 = ctx.xfer_stats()
    print(f"   get_h: wall {st['d2h_wall_s']*1e3:.2f} ms host copy {st['d2h_host_copy_s']*1e3:.2f} ms | again at once: {dt2*1e3:.2f} ms (wall {(st2['d2h_wall_s']-st['d2h_wall_s'])*1e3:.2f}, host copy {(st2['d2h_host_copy_s']-st['d2h_host_copy_s'])*1e3:.2f})")
    t.append(time.perf_counter())
    o = pl.get_objective(); lap()
    pl.close(); lap()
    names = ["create", "set_v", "set_w", "set_h", "init", "run", "get_w", "get_h", "get_h#2", "get_obj", "close"]
    d = np.diff(t) * 1e3
    print("steps ms: " + "  ".join(f"{n} {x:.2f}" for n, x in zip(names, d)) + f"   total {sum(d):.2f}")
    t0 = time.perf_counter(); sparse_nmf(Vh, pd, ctx=ctx); print(f"one-shot sparse_nmf call: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
