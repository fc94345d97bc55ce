"""Dev check (run on the GPU box): the pipelined kernels against the plain ones on awkward shapes, bit for bit.
  H-only, 2 iterations : k_hstep_rp                      vs  k_hstep            (SNMF_HSTEP_RP=0)
  full,   3 iterations : k_hstep_rp + k_wstats<..,NL=4>  vs  k_hstep + k_wstats<..,NL=0>  (SNMF_WSTATS_NL=0)
Shapes: a fixed list of known-nasty ones plus seeded random draws (F with and without the extra row, few / many row
and column tiles, one to several tiles per workgroup).  usage: python scripts/rp_shape_probe.py [n_random]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

FIXED = [(65, 70, 12000), (65, 70, 6000), (65, 128, 12000), (129, 70, 12000), (65, 32, 12000), (64, 70, 12000), (97, 96, 20000),
         (257, 40, 20000), (33, 8, 30000), (161, 200, 9000), (257, 256, 30000), (225, 100, 17000), (513, 64, 12000), (513, 200, 9000),
         (385, 100, 12000)]


def shapes(n_random):
    rs = np.random.RandomState(11)
    out = list(FIXED)
    for _ in range(n_random):
        nf = int(rs.randint(1, 9))
        F = 32 * nf + int(rs.choice([1, 0, -int(rs.randint(1, 31))]))
        F = max(F, 8)
        r = int(rs.randint(4, 300))
        if ((r + 31) // 32 * 32 + 4 + (F + 31) // 32 * 32 + 12) * 256 > 160 * 1024:
            r = 64
        T = int(rs.choice([900, 5000, 9000, 17000, 33000]))
        out.append((F, r, T))
    return out


if len(sys.argv) > 2 and sys.argv[1] == "child":
    from se_snmf_nat_amd import Context, Plan
    ctx = Context(0)
    tag, nrand = sys.argv[2], int(sys.argv[3])
    for F, r, T in shapes(nrand):
        rs = np.random.default_rng(F * 1000 + r)
        V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
        W0 = rs.random((F, r)); H0 = rs.random((r, T)).astype(np.float32)
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=2, conv_eps=0.0, cost_check=True, sparsity=1.0, w_update_ind=np.zeros(r, bool))
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
        h = pl.get_h(np.float32); pl.close()
        pl = Plan(ctx, F, T, r, beta=1.0, max_iter=3, conv_eps=0.0, cost_check=True, sparsity=1.0)
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
        w = pl.get_w(); geo = pl.describe(); pl.close()
        np.savez(f"/tmp/rp_probe_{tag}_{F}_{r}_{T}.npz", h=h, w=w, geo=geo)
    sys.exit(0)

nrand = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for tag, env in (("new", {}), ("old", dict(SNMF_HSTEP_RP="0", SNMF_WSTATS_NL="0"))):
    subprocess.run([sys.executable, __file__, "child", tag, str(nrand)], env=dict(os.environ, **env), check=True)
nbad = 0
for F, r, T in shapes(nrand):
    a = np.load(f"/tmp/rp_probe_new_{F}_{r}_{T}.npz"); b = np.load(f"/tmp/rp_probe_old_{F}_{r}_{T}.npz")
    dh = np.abs(a["h"] - b["h"]).max() / np.abs(b["h"]).max()
    dw = np.abs(a["w"] - b["w"]).max() / np.abs(b["w"]).max()
    rp = "k_hstep_rp" in str(a["geo"])
    ok = dh == 0 and dw < 1e-6
    nbad += not ok
    print(f"F={F} r={r} T={T} rp={int(rp)}: H-only max|dH|/max|H| = {dh:.1e}   full max|dW|/max|W| = {dw:.1e}  {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", nbad)
sys.exit(1 if nbad else 0)
