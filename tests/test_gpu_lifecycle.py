"""Lifecycle of the C-ABI objects on the device: plans, frame-stream plans and online separators give back every
byte they allocated (a basis-training driver creates and destroys one plan per solve, run_basis_DNMF.m:36-55; a
separation service one separator per stream), and a context survives its plans."""
import os

import numpy as np
import pytest

from oracle.sparse_nmf_oracle import synth_problem

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _one_cycle(ctx, with_online):
    from se_snmf_nat_amd import Plan
    V, W0, H0 = synth_problem(257, 6000, 64)
    for kw in (dict(), dict(w_update_ind=np.zeros(64, bool)), dict(h_update_ind=np.zeros(64, bool)), dict(beta=2.0)):
        a = dict(beta=1.0, max_iter=3, conv_eps=0.0, cost_check=True, sparsity=5.0)
        a.update(kw)
        pl = Plan(ctx, 257, 6000, 64, **a)
        pl.set_v(V.astype(np.float32))
        pl.set_w(W0)
        pl.set_h(H0.astype(np.float32))
        pl.init()
        pl.run()
        pl.get_w(np.float32)
        pl.close()
    if with_online:
        from se_snmf_nat_amd.online import OnlineSeparator, default_settings
        B = np.load(os.path.join(GOLD, "ref_data.npz"))["B"].astype(np.float64)
        s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"][:12 * 160]
        rs = np.random.RandomState(1)
        Bx, Bd, H0o, Ad0 = B[:, :100], B[:, 100:], rs.random_sample(200), rs.random_sample((50, 100))
        for adapt in (1, 0):
            ps = default_settings()
            if not adapt:
                ps["adapt_train_N"] = 0
            sep = OnlineSeparator(Bx, Bd, ps, H0=H0o, Ad_blk0=Ad0, ctx=ctx)
            sep.process(s, flush=True)
            sep.close()


def test_repeated_create_destroy_returns_all_device_memory(gpu_ctx):
    _one_cycle(gpu_ctx, True)  # first use: code objects, torch's own context, allocator pools
    gpu_ctx.sync()
    before = _free_bytes()
    for _ in range(15):
        _one_cycle(gpu_ctx, True)
    gpu_ctx.sync()
    after = _free_bytes()
    # the driver hands memory out in 2 MiB granules; anything that leaks per cycle shows up 15-fold
    assert before - after < (8 << 20), f"device memory shrank by {(before - after) / 2**20:.1f} MiB over 15 cycles"


def test_context_outlives_its_plans_and_results_do_not_depend_on_history(gpu_ctx):
    from se_snmf_nat_amd import sparse_nmf
    V, W0, H0 = synth_problem(129, 500, 24)
    p = dict(cf="kl", sparsity=0.5, max_iter=12, conv_eps=0.0, init_w=W0, init_h=H0, cost_check=1)
    w0, h0, o0 = sparse_nmf(V, p, ctx=gpu_ctx)
    _one_cycle(gpu_ctx, False)
    w1, h1, o1 = sparse_nmf(V, p, ctx=gpu_ctx)
    assert np.array_equal(w0, w1) and np.array_equal(h0, h1) and np.array_equal(o0["cost"], o1["cost"])
