"""snmf_plan_run's fused launches: the W finish (k_wfin: the chunk reduction and the W update in one launch) against the step API's two
launches (k_reduce -> statistics in memory -> k_wapply), which is the path the statistics take when they are summed over
ranks in between (se_snmf_nat_amd/dist.py, csrc/snmf_multi.h).  Both add the chunk slabs in the same fixed order, so W, H,
the objective and the stop index must agree bit for bit -- this is what lets the sharded tests pin the sharding algebra
against the single-device solve.  (src/sparse_nmf.m:215-244 is the update both implement.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "kl_full": dict(F=257, T=3000, r=40, beta=1.0, iters=12),
    "kl_full_513_r100": dict(F=513, T=9000, r=100, beta=1.0, iters=6),
    "kl_wonly": dict(F=129, T=2500, r=24, beta=1.0, iters=10, h_ind=False),
    "kl_semi": dict(F=64, T=900, r=48, beta=1.0, iters=10, w_part=16),
    "kl_early_stop": dict(F=129, T=2000, r=20, beta=1.0, iters=60, conv_eps=2e-3),
    "ed_full": dict(F=257, T=3000, r=40, beta=2.0, iters=10),
    "ed_r300": dict(F=200, T=2500, r=300, beta=2.0, iters=5),
    "is_full": dict(F=129, T=2000, r=16, beta=0.0, iters=8),
    "b05_nocost": dict(F=97, T=1500, r=12, beta=0.5, iters=8, cost_check=False),
    # H-only: the run loop folds the objective and tests convergence in one launch (step API: k_reduce + k_check)
    "kl_honly_513": dict(F=513, T=9000, r=200, beta=1.0, iters=6, w_none=True),
    "kl_honly_early_stop": dict(F=129, T=2000, r=20, beta=1.0, iters=80, w_none=True, conv_eps=2e-3),
    "ed_honly": dict(F=257, T=3000, r=40, beta=2.0, iters=8, w_none=True),
    # ... which since round 6 is no launch at all: the H step's LAST workgroup to arrive folds the partials in k_reduce's order and runs the
    # test (obj_partial_out in csrc/snmf_kernels.h).  One case per H-step kernel family, several tiles per workgroup:
    "kl_honly_rp_r256": dict(F=257, T=20000, r=256, beta=1.0, iters=5, w_none=True, kernel="k_hstep_rp"),
    "kl_honly_rh_r100": dict(F=513, T=12000, r=100, beta=1.0, iters=5, w_none=True, kernel="k_hstep_rh"),
    "kl_honly_sf_r200": dict(F=64, T=12000, r=200, beta=1.0, iters=6, w_none=True, kernel="k_hstep_sf"),
    "kl_honly_sf_early_stop": dict(F=64, T=12000, r=100, beta=1.0, iters=80, w_none=True, conv_eps=2e-3, kernel="k_hstep_sf"),
    "kl_honly_sr_r32": dict(F=257, T=12000, r=32, beta=1.0, iters=6, w_none=True, kernel="k_hstep_sr"),
    "kl_honly_sr_early_stop": dict(F=513, T=10000, r=20, beta=1.0, iters=80, w_none=True, conv_eps=2e-3, kernel="k_hstep_sr"),
    "is_honly": dict(F=129, T=9000, r=16, beta=0.0, iters=6, w_none=True),
    # the online adaptation's solve (src/bnmf_sep_event_RT_IS16.m:330-335): W-only, some columns fixed, m_a = 100 frames
    "ed_adapt_513x100": dict(F=513, T=100, r=50, beta=2.0, iters=40, h_ind=False, w_part=10, conv_eps=1e-4),
    "kl_adapt_513x100": dict(F=513, T=100, r=50, beta=1.0, iters=40, h_ind=False, w_part=10, conv_eps=1e-4),
}


def _mk(gpu_ctx, c):
    from se_snmf_nat_amd import Plan
    F, T, r = c["F"], c["T"], c["r"]
    rs = np.random.default_rng(F + T + r)
    V = (rs.gamma(0.5, 1.0, (F, 8)) @ rs.gamma(0.3, 1.0, (8, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    kw = {}
    if c.get("h_ind") is False:
        kw["h_update_ind"] = np.zeros(r, bool)
    if c.get("w_none"):
        kw["w_update_ind"] = np.zeros(r, bool)
    if "w_part" in c:
        wi = np.zeros(r, bool); wi[c["w_part"]:] = True
        kw["w_update_ind"] = wi
    pl = Plan(gpu_ctx, F, T, r, beta=c["beta"], max_iter=c["iters"], conv_eps=c.get("conv_eps", 0.0),
              cost_check=c.get("cost_check", True), sparsity=2.0, **kw)
    pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init()
    return pl


@pytest.mark.parametrize("name", list(CASES))
def test_fused_w_finish_equals_reduce_then_apply(gpu_ctx, name):
    import torch
    c = CASES[name]
    a = _mk(gpu_ctx, c)
    assert ("W finish (run loop): none (objective fold + convergence test on the H step's last workgroup)" if c.get("w_none") else "W finish (run loop): k_wfin") in a.describe()
    if "kernel" in c:
        assert c["kernel"] in a.describe()
    a.run()
    Wa, Ha = a.get_w(), a.get_h()
    div_a, cost_a, n_a = a.get_objective()
    a.close()
    b = _mk(gpu_ctx, c)
    stats = torch.zeros(b.stats_len(), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(c["iters"]):
        b.hstep(); b.wstats(stats.data_ptr()); b.wapply(stats.data_ptr())
    if c.get("cost_check", True):
        b.objstats(stats.data_ptr()); b.objapply(stats.data_ptr())
    gpu_ctx.sync()
    Wb, Hb = b.get_w(), b.get_h()
    div_b, cost_b, n_b = b.get_objective()
    b.close()
    assert n_a == n_b
    if name.endswith("early_stop"):
        assert 1 < n_a < c["iters"]  # the case does stop early
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
    assert np.array_equal(cost_a, cost_b) and np.array_equal(div_a, div_b)


@pytest.mark.parametrize("shape", [(257, 9000, 256, 1.0), (513, 9000, 100, 1.0), (300, 1500, 300, 2.0)], ids=lambda s: "F%d_T%d_r%d_b%g" % s)
def test_run_in_pieces_and_plan_reuse(gpu_ctx, shape):
    """snmf_plan_run(n) twice is snmf_plan_run(2n) once, and a second solve on the same plan (set_w / set_h / init again)
    repeats the first bit for bit: nothing a launch leaves behind -- the split tiles' arrival counters, the slabs of the
    Gram form, the extra-row sums in LDS -- may leak into the next one."""
    from se_snmf_nat_amd import Plan
    F, T, r, beta = shape
    rs = np.random.default_rng(F + r)
    V = (rs.gamma(0.5, 1.0, (F, 8)) @ rs.gamma(0.3, 1.0, (8, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    pl = Plan(gpu_ctx, F, T, r, beta=beta, max_iter=10, conv_eps=0.0, cost_check=True, sparsity=2.0)
    pl.set_v(V)
    res = []
    for pieces in ((10,), (3, 4, 3), (10,)):
        pl.set_w(W0); pl.set_h(H0); pl.init()
        for n in pieces:
            pl.run(n)
        res.append((pl.get_w(), pl.get_h(), pl.get_objective()[1].copy()))
    pl.close()
    for w, h, c in res[1:]:
        assert np.array_equal(w, res[0][0]) and np.array_equal(h, res[0][1]) and np.array_equal(c, res[0][2])


@pytest.mark.parametrize("shape", [(257, 20000, 256), (513, 12000, 200), (64, 12000, 100), (257, 12000, 32), (129, 1500, 24)], ids=lambda s: "F%d_T%d_r%d" % s)
def test_h_only_objective_fold_on_the_h_step_equals_the_reduce_launch(gpu_ctx, shape, monkeypatch):
    """The H-only run loop (the reference's enhancement stage: run_basis_DNMF.m:40, src/sparse_nmf.m:186-208 with W fixed) folds
    the objective and tests convergence on the H step itself; SNMF_HFOLD=0 keeps round 5's k_reduce launch per iteration.  Same
    order of additions: H, both histories and the stop index agree in every bit -- also when the solve runs in pieces, stops
    early, or the plan is reused (the arrival counter is monotonic and must stay aligned)."""
    from se_snmf_nat_amd import Plan
    F, T, r = shape
    rs = np.random.default_rng(F + T + r)
    V = (rs.gamma(0.5, 1.0, (F, 8)) @ rs.gamma(0.3, 1.0, (8, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    res = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("SNMF_HFOLD", fold)
        out = []
        for eps, pieces in ((0.0, (12,)), (0.0, (1, 4, 2, 5)), (3e-3, (40,)), (0.0, (12,))):
            pl = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=sum(pieces), conv_eps=eps, cost_check=True, sparsity=1.0, w_update_ind=np.zeros(r, bool))
            assert ("on the H step's last workgroup" in pl.describe()) == (fold == "1")
            pl.set_v(V)
            for rep in range(2):  # (the second solve on the same plan repeats the first)
                pl.set_w(W0); pl.set_h(H0); pl.init()
                for n in pieces:
                    pl.run(n)
                div, cost, n_it = pl.get_objective()
                out.append((pl.get_h(np.float32), div.copy(), cost.copy(), n_it))
            pl.close()
        res[fold] = out
    for a, b in zip(res["1"], res["0"]):
        assert a[3] == b[3]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert 1 < res["1"][4][3] < 40  # the early-stop leg does stop early
    for k in (1, 2, 3, 6, 7):  # pieces == one run == reuse
        assert np.array_equal(res["1"][0][0], res["1"][k][0]) and np.array_equal(res["1"][0][2], res["1"][k][2])
