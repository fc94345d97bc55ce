"""Dev check: 64-bit indexing at large T (257 x 4,000,000, r = 256: V 4.1 GB, H 2 x 4.1 GB on the device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from se_snmf_nat_amd import Context, Plan
F, T, r = 257, int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 256
rs = np.random.default_rng(0)
t = time.time()
blk = 100_000
V = np.empty((F, T), dtype=np.float32, order="F")
Wt = rs.gamma(0.5, 1.0, (F, 32)).astype(np.float32)
for t0 in range(0, T, blk):
    n = min(blk, T - t0)
    V[:, t0:t0 + n] = Wt @ rs.gamma(0.3, 1.0, (32, n)).astype(np.float32) + np.float32(1e-3)
H0 = rs.random((r, T), dtype=np.float32)
W0 = rs.random((F, r), dtype=np.float32)
print("host data %.1fs" % (time.time() - t), flush=True)
ctx = Context(0)
plan = Plan(ctx, F, T, r, beta=1.0, max_iter=4, conv_eps=0.0, cost_check=True, sparsity=5.0)
t = time.time(); plan.set_v(V); plan.set_w(W0); plan.set_h(H0); plan.init(); print("upload %.1fs" % (time.time() - t), flush=True)
t = time.time(); n = plan.run(); ctx.sync(); dt = time.time() - t
div, cost, ni = plan.get_objective()
w = plan.get_w(np.float32)
h_tail = plan.get_h(np.float32)[:, -5:]
print("iterations", n, "%.1f ms/iteration" % (dt / 4 * 1e3), "cost", cost[:4], "monotone", bool(np.all(np.diff(cost[:4]) <= 0)))
print("W unit norm", float(np.abs(np.sqrt((w.astype(np.float64) ** 2).sum(0)) - 1).max()), "H tail finite/non-negative", bool(np.isfinite(h_tail).all() and (h_tail >= 0).all()))
# frame locality at the far end of the index range
hp = dict(beta=1.0, max_iter=2, conv_eps=0.0, cost_check=True, sparsity=5.0, w_update_ind=np.zeros(r, bool))
plan.close()
full = Plan(ctx, F, T, r, **hp); full.set_v(V); full.set_w(W0); full.set_h(H0); full.init(); full.run()
hf = full.get_h(np.float32)[:, T - 7000:T - 1000].copy(); full.close()
part = Plan(ctx, F, 6000, r, **hp); part.set_v(np.ascontiguousarray(V[:, T - 7000:T - 1000])); part.set_w(W0); part.set_h(np.ascontiguousarray(H0[:, T - 7000:T - 1000])); part.init(); part.run()
hp_ = part.get_h(np.float32)
print("frame locality at the end of the range: rel", float(np.linalg.norm(hp_ - hf) / np.linalg.norm(hf)))
