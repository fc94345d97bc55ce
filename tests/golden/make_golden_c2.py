"""Full-size golden for BASELINE configs[1] (C2: 257 x 100000, r = 256, KL, sparsity 5) -- VERDICT r1 #5.

Runs the fp64 oracle (oracle/sparse_nmf_oracle.py, restating src/sparse_nmf.m:71-292) on EXACTLY the inputs bench.py
times (bench.make_problem, V and H0 rounded to fp32 as the device receives them, W0 fp64) and stores

    cost[0:N_COST], div[0:N_COST]   objective after iterations 1..N_COST   (src/sparse_nmf.m:248-264)
    W12                             the dictionary after 12 iterations      (257 x 256, fp64)
    H12_head / H12_tail             the first / last 64 frames of H after 12 iterations
    cost12 / div12                  the 12-iteration objective vectors (equal to cost[0:12]; kept as a self check)

as tests/golden/c2_full_257x100000_r256.npz (about 0.6 MB).  bench.py compares its `final_cost` with cost[n_iter-1]
("cost_vs_oracle"); tests/test_gpu_parity.py::test_full_size_c2_against_the_oracle_golden compares W, H and every
cost of a 12-iteration device solve.  About 2.6 s per iteration on 8 vCPU: ~12 minutes for N_COST = 260.

Run from the repo root:  python tests/golden/make_golden_c2.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N_COST = int(os.environ.get("SNMF_C2_NCOST", "260"))

if __name__ == "__main__":
    from bench import F_, R_, SPARSITY, T_, make_problem
    from oracle.sparse_nmf_oracle import sparse_nmf
    V, W0, H0 = make_problem(F_, T_, R_)
    V = V.astype(np.float32).astype(np.float64)
    H0 = H0.astype(np.float32).astype(np.float64)
    base = dict(cf="kl", sparsity=SPARSITY, conv_eps=0, init_w=W0, init_h=H0, cost_check=1)
    t = time.time()
    w12, h12, o12 = sparse_nmf(V, dict(base, max_iter=12))
    print(f"12 iterations: {time.time() - t:.0f} s, cost {o12['cost'][-1]:.6f}", flush=True)
    t = time.time()
    _, _, o = sparse_nmf(V, dict(base, max_iter=N_COST))
    print(f"{N_COST} iterations: {time.time() - t:.0f} s, cost {o['cost'][-1]:.6f}", flush=True)
    assert np.array_equal(o["cost"][:12], o12["cost"])
    dst = os.path.join(ROOT, "tests", "golden", "c2_full_257x100000_r256.npz")
    np.savez_compressed(dst, cost=o["cost"], div=o["div"], W12=w12, H12_head=h12[:, :64], H12_tail=h12[:, -64:],
                        cost12=o12["cost"], div12=o12["div"], F=F_, T=T_, r=R_, sparsity=SPARSITY)
    print(dst, os.path.getsize(dst), "bytes")
