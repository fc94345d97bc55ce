"""How the headline kernels' time moves over the first iterations of a solve from a random start (C2 size): HIP-event averages of
the H step and the W statistics per block of 10 iterations.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from se_snmf_nat_amd import Context, Plan

ctx = Context(0)
F, T, r = 257, 100000, 256
V, W0, H0 = bench.make_problem(F, T, r)
for rep in range(2):
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=400, conv_eps=0.0, cost_check=True, sparsity=5.0)
    pl.set_v(V.astype(np.float32)); pl.set_w(W0); pl.set_h(H0.astype(np.float32)); pl.init()
    row = []
    for blk in range(24):
        ctx.timing(True)
        pl.run_async(10); ctx.sync()
        h, w = ctx.timing_get("hstep")[0], ctx.timing_get("wstats")[0]
        ctx.timing(False)
        row.append((h * 1e3, w * 1e3))
    print("rep", rep, " ".join(f"{a:.0f}/{b:.0f}" for a, b in row), flush=True)
    pl.close()
