"""The drop-in leg of bench.py in isolation: sparse_nmf(v, p) on host fp64 arrays, C2 size, K iterations, several calls in a row --
call wall, the library's own upload / copy-out walls (ctx.xfer_stats) per call.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from se_snmf_nat_amd import Context, sparse_nmf

ctx = Context(0)
F, T, r = 257, 100000, 256
V, W0, H0 = bench.make_problem(F, T, r)
Vh, Wh, Hh = (np.asfortranarray(M, dtype=np.float64) for M in (V, W0, H0))
for K in (50, 200, 50):
    pd = dict(cf="kl", sparsity=5.0, max_iter=K, conv_eps=0, cost_check=1, init_w=Wh, init_h=Hh)
    sparse_nmf(Vh[:, :4096], dict(pd, init_h=Hh[:, :4096], max_iter=2), ctx=ctx)
    for i in range(5):
        ctx.xfer_stats(reset=True)
        t = time.perf_counter()
        w, h, o = sparse_nmf(Vh, pd, ctx=ctx)
        dt = time.perf_counter() - t
        st = ctx.xfer_stats()
        print(f"K={K} call {i}: {dt*1e3:7.2f} ms  h2d {st['h2d_wall_s']*1e3:6.2f} ms  d2h {st['d2h_wall_s']*1e3:6.2f} ms ({st['d2h_bytes']/1e6/max(st['d2h_wall_s'],1e-9)/1e3:5.1f} GB/s)  d2h host part {st.get('d2h_host_s', 0)*1e3:6.2f} ms", flush=True)
        del w, h
