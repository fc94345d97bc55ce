"""Development helper: run a set of parity cases on the GPU and print the errors vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.sparse_nmf_oracle import sparse_nmf as onmf, synth_problem
from se_snmf_nat_amd import sparse_nmf, SnmfError

def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

def case(name, F, T, r, p, wi=None, hi=None, scale="unit"):
    V, W0, H0 = synth_problem(F, T, r, scale=scale)
    p = dict(p, init_w=W0, init_h=H0, cost_check=p.get("cost_check", 1))
    if wi is not None: p["w_update_ind"] = wi
    if hi is not None: p["h_update_ind"] = hi
    t = time.time()
    try:
        w, h, o = sparse_nmf(V, p)
    except SnmfError as e:
        print(f"{name:34s} ERROR {e}"); return
    tg = time.time() - t
    wr, hr, orf = onmf(V, p)
    n = min(len(o["cost"]), len(orf["cost"]))
    ec = np.max(np.abs(o["cost"][:n] - orf["cost"][:n]) / np.abs(orf["cost"][:n])) if n else 0
    print(f"{name:34s} it {o['n_iter']:3d}/{orf['n_iter']:3d} relW {rel(w,wr):.2e} relH {rel(h,hr):.2e} "
          f"maxrelcost {ec:.2e} nan {np.isnan(w).any() or np.isnan(h).any()} t={tg:.2f}s")

kl = dict(cf="kl", sparsity=5, max_iter=20, conv_eps=0)
case("kl full 64x96 r32", 64, 96, 32, kl)
case("kl full 257x640 r40", 257, 640, 40, kl)
case("kl H-only 257x640 r40", 257, 640, 40, kl, wi=np.zeros(40, bool))
case("kl W-only 257x640 r40", 257, 640, 40, kl, hi=np.zeros(40, bool))
case("kl semi 257x640 r40", 257, 640, 40, kl, wi=np.arange(40) >= 20)
case("kl full 513x100 r50", 513, 100, 50, kl)
case("kl full T=1 513 r200 Honly", 513, 1, 200, dict(kl, max_iter=100, conv_eps=1e-3), wi=np.zeros(200, bool))
case("kl earlystop 257x2000 r40", 257, 2000, 40, dict(kl, max_iter=100, conv_eps=1e-3))
case("kl power 257x2000 r40", 257, 2000, 40, dict(kl, max_iter=30), scale="power")
case("kl big-ish 257x20000 r256", 257, 20000, 256, dict(kl, max_iter=5))
case("ed full 257x640 r40", 257, 640, 40, dict(cf="ed", sparsity=5, max_iter=20))
case("ed W-only", 257, 640, 40, dict(cf="ed", sparsity=5, max_iter=20), hi=np.zeros(40, bool))
case("ed H-only", 257, 640, 40, dict(cf="ed", sparsity=5, max_iter=20), wi=np.zeros(40, bool))
case("is full 257x640 r40", 257, 640, 40, dict(cf="is", sparsity=0.1, max_iter=20))
case("b0.5 full 257x640 r40", 257, 640, 40, dict(cf="x", beta=0.5, sparsity=1, max_iter=20))
case("b1.5 full 129x300 r300", 129, 300, 300, dict(cf="x", beta=1.5, sparsity=1, max_iter=10))
case("kl r=600 H-only", 100, 200, 600, dict(kl, max_iter=5), wi=np.zeros(600, bool))
case("kl r=600 full", 100, 200, 600, dict(kl, max_iter=5))
case("kl no cost_check", 257, 640, 40, dict(kl, cost_check=0))
case("kl rvec sparsity", 129, 300, 24, dict(kl, sparsity=np.linspace(0, 9, 24)))
case("kl full-matrix sparsity", 129, 300, 24, dict(kl, sparsity=np.abs(np.random.RandomState(3).randn(24, 300)) * 4))
