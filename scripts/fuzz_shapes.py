"""Development helper: random shapes / modes through the product path against the fp64 oracle (GPU box).
   python scripts/fuzz_shapes.py [seed] [n_cases] [time budget s] [focus]  -> one line per case, FAIL lines at the end.
focus = "r5": only the envelopes of round 5's kernels -- k_iter_sf (F = 33..64, r = 65..128, KL, both factors updated) and
k_hstep_rp<., CUT> (>= 4 row tiles, r <= 64, more tiles than CUs) -- with their edge cases (partial last tile, F on both sides of
32n + 1, every sparsity form, cost on / off).  focus = "big": 33000..90000 frames (4..11 tiles per workgroup) on the main families.
Shapes are drawn to land on the plan's geometry switches: tile counts around multiples of the CU count (the split last
round), F on both sides of 32n+1, r around the 32-column tiles and the LX / NK limits, all divergences and update modes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.sparse_nmf_oracle import sparse_nmf as onmf
from se_snmf_nat_amd import SnmfError, sparse_nmf

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 420.0
focus = sys.argv[4] if len(sys.argv) > 4 else ""
rs = np.random.default_rng(seed)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def draw_r5():
    if rs.integers(0, 2) == 0:   # the fused small-F iteration
        F = int(rs.choice([64, 64, 40, 33, 63, int(rs.integers(33, 65))]))
        r = int(rs.choice([100, 128, 65, 96, 97, int(rs.integers(65, 129))]))
        T = int(rs.choice([int(rs.integers(1, 300)), int(rs.integers(300, 9000)), int(8192 + 32 * rs.integers(1, 700) + rs.integers(-31, 1)),
                           int(32 * 256 * rs.integers(1, 4) + rs.integers(-40, 40))]))
        mode = str(rs.choice(["full", "full", "full", "semi"]))
    else:                        # the contraction cut
        F = int(rs.choice([257, 513, 129, 128, 512, 385, 256, int(rs.integers(128, 514))]))
        r = int(rs.choice([20, 10, 30, 32, 33, 50, 64, 1, int(rs.integers(1, 65))]))
        T = int(8192 + 32 * rs.integers(1, 500) + rs.integers(-31, 1))
        mode = str(rs.choice(["full", "h", "h", "semi"]))
    sp = str(rs.choice(["scalar", "scalar", "vec", "mat", "zero"]))
    return F, T, r, 1.0, mode, sp


def draw_big():
    """long solves: 4..11 tiles of 32 frames per workgroup on 256 CUs -- every buffer of every pipeline wraps around several times"""
    F = int(rs.choice([257, 513, 129, 64, 64, 385, 512, 40, int(rs.integers(33, 65)), int(rs.integers(100, 514))]))
    r = int(rs.choice([20, 30, 32, 50, 64, 100, 100, 128, 200, 256, int(rs.integers(1, 257))]))
    T = int(rs.integers(33000, 90000))
    beta = float(rs.choice([1.0, 1.0, 1.0, 1.0, 2.0, 0.0, 0.5]))
    mode = str(rs.choice(["full", "full", "h", "w", "semi"]))
    sp = str(rs.choice(["scalar", "scalar", "vec", "zero"]))
    return F, T, r, beta, mode, sp


def draw():
    if focus == "r5":
        return draw_r5()
    if focus == "big":
        return draw_big()
    F = int(rs.choice([257, 513, 129, 65, 64, 128, 512, 385, 97, 64, 40, 32, int(rs.integers(8, 65)), int(rs.integers(8, 600)), int(rs.integers(8, 200))]))
    r = int(rs.choice([int(rs.integers(1, 40)), int(rs.integers(90, 132)), int(rs.integers(190, 260)), 100, 200, 256, 40,
                       int(rs.integers(257, 700)), int(rs.integers(1, 300))]))
    kind = rs.integers(0, 4)
    if kind == 0:
        T = int(rs.integers(1, 400))
    elif kind == 1:
        T = int(8192 * rs.integers(1, 4) + rs.integers(-3000, 3000))
    elif kind == 2:
        T = int(rs.integers(400, 9000))
    else:
        T = int(8192 + 32 * rs.integers(1, 140) + rs.integers(-31, 1))
    big = rs.integers(0, 12)  # now and then a shape beyond the fused kernels' envelope (csrc/snmf_generic.h)
    if big == 0:
        F, r, T = int(rs.integers(2500, 3200)), int(rs.integers(1, 80)), int(rs.integers(1, 700))
    elif big == 1:
        F, r, T = int(rs.choice([129, 200, 257, 64])), int(rs.integers(1030, 1300)), int(rs.integers(1, 5000))
    while F * T * r > 6e9 or ((F + r) > 2400 and big > 1):
        T = max(1, T // 2)
        if (F + r) > 2400:
            r = r // 2
    beta = float(rs.choice([1.0, 1.0, 1.0, 2.0, 0.0, 0.5, 1.5]))
    mode = str(rs.choice(["full", "full", "h", "w", "semi"]))
    sp = str(rs.choice(["scalar", "scalar", "vec", "mat", "zero"]))
    return F, T, r, beta, mode, sp


fails = []
t_start = time.time()
for ci in range(n_cases):
    if time.time() - t_start > budget:
        print(f"(time budget reached after {ci} cases)")
        break
    F, T, r, beta, mode, sp = draw()
    V = (rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3)
    dv = int(rs.integers(0, 8))  # data variants: spectrogram-like dynamic range, global scale, silent rows / frames
    if dv == 0:
        V = V ** 3 * 1e3
    elif dv == 1:
        V = V * float(rs.choice([1e-5, 1e5]))
    elif dv == 2:
        V[rs.random(F) < 0.1, :] = 0.0
        V[:, rs.random(T) < 0.05] = 0.0
    W0 = rs.random((F, r))
    H0 = rs.random((r, T))
    iters = int(rs.integers(2, 6))
    # early stop (src/sparse_nmf.m:272-284): now and then a solve that may stop by itself -- the stop index must be the oracle's unless the
    # oracle's own decision was within 2 % of the threshold at some iteration (then the case is counted as borderline, not compared)
    eps = float(rs.choice([0, 0, 0, 1e-3, 3e-3, 1e-2])) if focus != "big" else 0.0
    if eps > 0:
        iters = int(rs.integers(8, 40))
    p = dict(cf={1.0: "kl", 2.0: "ed", 0.0: "is"}.get(beta, "x"), beta=beta, max_iter=iters, conv_eps=eps,
             cost_check=1 if eps > 0 else int(rs.integers(0, 4) > 0), init_w=W0, init_h=H0)
    p["sparsity"] = {"scalar": float(rs.choice([0.1, 1.0, 5.0])), "zero": 0.0, "vec": rs.random(r) * 4,
                     "mat": rs.random((r, T)) * 3}[sp]
    if mode == "h":
        p["w_update_ind"] = np.zeros(r, bool)
    elif mode == "w":
        p["h_update_ind"] = np.zeros(r, bool)
    elif mode == "semi":
        p["w_update_ind"] = np.arange(r) >= r // 2
    tag = f"F={F} T={T} r={r} beta={beta} {mode} sp={sp} it={iters} eps={eps:g} cc={p['cost_check']} dv={dv}"
    try:
        w, h, o = sparse_nmf(V, p)
    except SnmfError as e:
        print(f"{ci:3d} {tag}: REFUSED {str(e)[:90]}")
        continue
    wr, hr, orf = onmf(V, p)
    if eps > 0:
        c = np.asarray(orf["cost"], float)
        rc = np.abs(np.diff(c)) / np.abs(c[:-1]) if len(c) > 1 else np.array([])
        if len(rc) and np.min(np.abs(rc - eps)) < 0.02 * eps:
            print(f"{ci:3d} {tag}: borderline stop decision in the oracle itself (skipped)")
            continue
        if o["n_iter"] != orf["n_iter"]:
            print(f"{ci:3d} {tag}: n_iter {o['n_iter']} != oracle {orf['n_iter']}  <<< FAIL", flush=True)
            fails.append(tag + " (stop index)")
            continue
    n = min(len(o["cost"]), len(orf["cost"]))
    ec = float(np.max(np.abs(o["cost"][:n] - orf["cost"][:n]) / np.abs(orf["cost"][:n]))) if n and p["cost_check"] else 0.0
    ew, eh = rel(w, wr), rel(h, hr)
    bad = (not np.isfinite(w).all()) or (not np.isfinite(h).all()) or ew > 2e-4 or eh > 2e-4 or ec > 2e-5
    print(f"{ci:3d} {tag}: relW {ew:.1e} relH {eh:.1e} cost {ec:.1e}{'  <<< FAIL' if bad else ''}", flush=True)
    if bad:
        fails.append(tag)
print(f"{len(fails)} failures")
for f in fails:
    print("FAIL", f)
