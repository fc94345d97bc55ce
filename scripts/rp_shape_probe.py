"""Dev check: k_hstep_rp against k_hstep (SNMF_HSTEP_RP=0) on awkward shapes, H-only, 2 iterations: max |dH| / max |H|."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

SHAPES = [(65, 70, 12000), (65, 70, 6000), (65, 128, 12000), (129, 70, 12000), (65, 32, 12000), (64, 70, 12000), (97, 96, 20000),
          (257, 40, 20000), (33, 8, 30000), (161, 200, 9000)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from se_snmf_nat_amd import Context, Plan
    ctx = Context(0)
    out = {}
    for F, r, T in SHAPES:
        rs = np.random.default_rng(F * 1000 + r)
        V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
        W0 = rs.random((F, r)); H0 = rs.random((r, T)).astype(np.float32)
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=2, conv_eps=0.0, cost_check=True, sparsity=1.0, w_update_ind=np.zeros(r, bool))
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
        h = pl.get_h(np.float32); pl.close()
        np.save(f"/tmp/rp_probe_{os.environ.get('SNMF_HSTEP_RP', '1')}_{F}_{r}_{T}.npy", h)
    sys.exit(0)

for rp in ("1", "0"):
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SNMF_HSTEP_RP=rp), check=True)
for F, r, T in SHAPES:
    a = np.load(f"/tmp/rp_probe_1_{F}_{r}_{T}.npy"); b = np.load(f"/tmp/rp_probe_0_{F}_{r}_{T}.npy")
    d = np.abs(a - b)
    bad = np.argwhere(d > 1e-4 * np.abs(b).max())
    print(f"F={F} r={r} T={T}: max|dH|/max|H| = {d.max() / np.abs(b).max():.2e}  bad entries {len(bad)}",
          ("first bad (k,t): " + str(bad[:4].tolist()) + " frames mod 32: " + str(sorted(set((bad[:, 1] % 32).tolist()))[:12]) + " tiles: " + str(sorted(set((bad[:, 1] // 32).tolist()))[:8])) if len(bad) else "")
