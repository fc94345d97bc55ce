#!/usr/bin/env python3
"""Microbenchmark of the single-frame solve kernel: 256 independent frames (one workgroup per CU), 100 fixed
iterations each -> microseconds per iteration of one workgroup.  Usage: python scripts/bench_frame.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from se_snmf_nat_amd import Context, Plan
ctx = Context(0)
ref = dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_data.npz")))
B, Y = ref["B"].astype(np.float64), ref["Y"].astype(np.float64)
H0 = np.random.RandomState(1).random_sample((200, 1)).astype(np.float32)
n, iters = 256, 100
Yl = np.asfortranarray(np.tile(Y, (1, 4))[:, :n], dtype=np.float32)
for cc in (True, False):
    plan = Plan(ctx, 513, n, 200, beta=1.0, max_iter=iters, conv_eps=0.0, cost_check=cc, sparsity=5.0, w_update_ind=np.zeros(200, bool))
    plan.set_w(B)
    plan.solve_frames(Yl, H0, dtype=np.float32)
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        plan.solve_frames(Yl, H0, dtype=np.float32)
    dt = (time.perf_counter() - t) / reps
    print("cost_check=%d: %.1f us per call, %.2f us per iteration (incl. ~host overhead/iters)" % (cc, dt * 1e6, dt * 1e6 / iters))
    plan.close()
