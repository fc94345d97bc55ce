"""Dev probe: per-iteration time in successive windows right after the GPU sat idle (does the clock ramp show?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import make_problem, SPARSITY
from se_snmf_nat_amd import Context, Plan
ctx = Context(0)
for Tn in (100_000, 12_500):
    V, W0, H0 = make_problem(257, 100_000, 256, 0, Tn)
    plan = Plan(ctx, 257, Tn, 256, beta=1.0, max_iter=5000, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
    plan.set_v(V.astype(np.float32)); plan.set_w(W0); plan.set_h(H0.astype(np.float32)); plan.init(); ctx.sync()
    for idle in (1.0, 0.05, 0.002):
        time.sleep(idle)
        out = []
        for w in range(12):
            n = 10 if w < 6 else 100
            t = time.perf_counter(); plan.run_async(n); ctx.sync(); out.append((n, (time.perf_counter() - t) / n * 1e3))
        print(f"T={Tn} after {idle}s idle:", " ".join(f"{n}x{ms:.4f}" for n, ms in out), flush=True)
    plan.close()
