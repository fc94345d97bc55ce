"""Multi-GPU entry behind the C ABI (include/snmf.h: snmf_multi_*, snmf_sparse_nmf_multi_*; SURVEY.md section 8b(1)
"multi-GPU variant taking a device list"): ONE process, the frame axis sharded over the ranks, a one-shot exchange of
the fp64 W statistics per iteration (src/sparse_nmf.m:215-239 reduced over T; every rank then applies the identical
epilogue :215-244 and convergence test :272-284).

A single-GPU test box cannot put the ranks on different devices, so the device list repeats device 0 (the ranks then
share it; the peer stores degenerate to local stores).  Everything else is the product path.  Checked against the fp64
oracle, against the unsharded solve, for bit-identical W replicas, and -- bit for bit -- against a hand-driven loop
over the plan step API whose exchange is a host-side fp64 sum in rank order.
"""
import ctypes as C

import numpy as np
import pytest

from oracle.sparse_nmf_oracle import sparse_nmf as oracle_nmf, synth_problem

pytestmark = pytest.mark.gpu
REL_WH = 1e-4
REL_COST = 1e-5


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


CASES = [
    dict(F=257, T=2100, r=64, cf="kl", sparsity=5.0, max_iter=15, conv_eps=0.0, n=2),
    dict(F=257, T=1500, r=40, cf="kl", sparsity=0.5, max_iter=80, conv_eps=2e-3, n=3),            # early stop, 3 ranks
    dict(F=129, T=700, r=24, cf="ed", sparsity=0.3, max_iter=10, conv_eps=0.0, n=2),
    dict(F=129, T=700, r=24, cf="is", sparsity=0.01, max_iter=10, conv_eps=0.0, n=4),
    dict(F=257, T=900, r=32, cf="kl", sparsity=5.0, max_iter=30, conv_eps=1e-3, n=2, w_ind=0),  # H-only: scalars only
    dict(F=257, T=900, r=32, cf="kl", sparsity=5.0, max_iter=30, conv_eps=1e-3, n=2, h_ind=0),  # W-only
]


def _p(case, W0, H0):
    p = dict(cf=case["cf"], sparsity=case["sparsity"], max_iter=case["max_iter"], conv_eps=case["conv_eps"], init_w=W0,
             init_h=H0, cost_check=1)
    if "w_ind" in case:
        p["w_update_ind"] = np.zeros(case["r"], bool)
    if "h_ind" in case:
        p["h_update_ind"] = np.zeros(case["r"], bool)
    return p


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"F{c['F']}-{c['cf']}-n{c['n']}-eps{c['conv_eps']}-"
                         f"{'Honly' if 'w_ind' in c else 'Wonly' if 'h_ind' in c else 'full'}")
def test_device_list_solve_matches_oracle_and_unsharded(gpu_ctx, case):
    from se_snmf_nat_amd import sparse_nmf
    V, W0, H0 = synth_problem(case["F"], case["T"], case["r"])
    p = _p(case, W0, H0)
    w, h, o = sparse_nmf(V, p, devices=[0] * case["n"])
    wr, hr, orf = oracle_nmf(V, p)
    assert o["n_iter"] == orf["n_iter"]
    assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH
    np.testing.assert_allclose(o["cost"], orf["cost"], rtol=REL_COST)
    w1, h1, o1 = sparse_nmf(V, p, ctx=gpu_ctx)
    assert o1["n_iter"] == o["n_iter"]
    assert rel(w, w1) < 1e-5 and rel(h, h1) < 1e-5  # only the summation order of the statistics differs


@pytest.mark.parametrize("mode,two_devices", [("events", False), ("flags", False), ("flags", True), ("events", True)],
                         ids=["events-one-device", "flags-one-device", "flags-two-devices", "events-two-devices"])
@pytest.mark.parametrize("unfused", [False, True], ids=["fused", "unfused"])
def test_handle_api_replicas_are_bit_identical_and_equal_the_step_api(gpu_ctx, lib, mode, two_devices, unfused, monkeypatch):
    """snmf_multi_* with 2 ranks against a hand-driven pair of plans over the step API (hstep -> wstats -> host-side fp64
    sum in rank order -> wapply): the one-shot exchange adds the slots in rank order too, so W, H and every cost must
    agree BIT FOR BIT; and the two W replicas of the multi handle must be identical.  Both orderings of the exchange
    (EVENTS: hipEvents + host barrier; FLAGS: arrival words polled on the device, the host only enqueues), on device 0
    twice (peer stores degenerate to local ones) and -- where the box has them -- on two PHYSICAL devices: peer-mapped
    fine-grained gather buffers, cross-device arrival words; the only check of the path on real peers."""
    import torch
    from se_snmf_nat_amd import Plan, _lib
    from se_snmf_nat_amd.api import _make_params
    if two_devices and lib.snmf_device_count() < 2:
        pytest.skip("needs two HIP devices")
    # round 4: the exchange rides on the iteration's own launches (k_reduce pushes into the peers' slots, k_wapply sums them:
    # four launches per iteration and rank instead of six); SNMF_MULTI_UNFUSED=1 keeps k_push_stats / k_sum_ranks -- same bits
    monkeypatch.setenv("SNMF_MULTI_UNFUSED", "1" if unfused else "0")
    F, T, r, iters = 257, 1300, 48, 9
    V, W0, H0 = synth_problem(F, T, r)
    V32, H32 = V.astype(np.float32), H0.astype(np.float32)
    cols = [0, 600, T]  # deliberately unbalanced ranges
    # ---- multi handle
    sp = _make_params(F, T, r, 1.0, iters, 0.0, 1, True, 0, 5.0, None, None)
    h = C.c_void_p()
    devs = np.asarray([0, 1 if two_devices else 0], np.int32)
    cb = np.asarray(cols, np.int64)
    _lib.check(lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 2, C.byref(sp), C.c_void_p(cb.ctypes.data), C.byref(h)))
    _lib.check(lib.snmf_multi_set_exchange(h, _lib.EXCHANGE_FLAGS if mode == "flags" else _lib.EXCHANGE_EVENTS))
    try:
        Vf, Hf, Wf = np.asfortranarray(V32), np.asfortranarray(H32), np.asfortranarray(W0)
        _lib.check(lib.snmf_multi_set_v_f32(h, C.c_void_p(Vf.ctypes.data), F))
        _lib.check(lib.snmf_multi_set_w_f64(h, C.c_void_p(Wf.ctypes.data), F))
        _lib.check(lib.snmf_multi_set_h_f32(h, C.c_void_p(Hf.ctypes.data), r))
        _lib.check(lib.snmf_multi_init(h))
        done = C.c_int32()
        _lib.check(lib.snmf_multi_run(h, iters, C.byref(done)))
        assert done.value == iters
        Wm = [np.empty((F, r), order="F") for _ in range(2)]
        for g in range(2):
            _lib.check(lib.snmf_multi_get_w_rank_f64(h, g, C.c_void_p(Wm[g].ctypes.data), F))
        Hm = np.empty((r, T), order="F")
        _lib.check(lib.snmf_multi_get_h_f64(h, C.c_void_p(Hm.ctypes.data), r))
        div, cost = np.zeros(iters), np.zeros(iters)
        n = C.c_int32()
        _lib.check(lib.snmf_multi_get_objective(h, C.c_void_p(div.ctypes.data), C.c_void_p(cost.ctypes.data), C.byref(n)))
    finally:
        lib.snmf_multi_destroy(h)
    assert np.array_equal(Wm[0], Wm[1])
    # ---- the same two shards driven by hand over the step API
    plans, stats = [], []
    for g in range(2):
        t0, t1 = cols[g], cols[g + 1]
        pl = Plan(gpu_ctx, F, t1 - t0, r, beta=1.0, max_iter=iters, conv_eps=0.0, cost_check=True, sparsity=5.0)
        pl.set_v(V32[:, t0:t1]); pl.set_w(W0); pl.set_h(H32[:, t0:t1]); pl.init()
        plans.append(pl)
        stats.append(torch.zeros(pl.stats_len(), dtype=torch.float64, device="cuda:0"))
    torch.cuda.synchronize()

    def exchange():
        gpu_ctx.sync()
        tot = stats[0].cpu().numpy() + stats[1].cpu().numpy()  # rank order, fp64
        for g in range(2):
            stats[g].copy_(torch.from_numpy(tot))
        torch.cuda.synchronize()

    for _ in range(iters):
        for g in range(2):
            plans[g].hstep(); plans[g].wstats(stats[g].data_ptr())
        exchange()
        for g in range(2):
            plans[g].wapply(stats[g].data_ptr())
    for g in range(2):
        plans[g].objstats(stats[g].data_ptr())
    exchange()
    for g in range(2):
        plans[g].objapply(stats[g].data_ptr())
    Ws = plans[0].get_w()
    Hs = np.concatenate([plans[0].get_h(), plans[1].get_h()], axis=1)
    _, cost_s, n_s = plans[0].get_objective()
    for pl in plans:
        pl.close()
    assert n_s == n.value == iters
    assert np.array_equal(Ws, Wm[0]) and np.array_equal(Hs, Hm) and np.array_equal(cost_s[:iters], cost)


def test_dnmf_three_solve_loop_over_a_device_list(gpu_ctx):
    """run_basis_DNMF.m:36-55 with every solve sharded over two ranks (BASELINE config 4 behind the reference's call)."""
    from se_snmf_nat_amd import run_basis_dnmf
    rs = np.random.default_rng(3)
    F, T, Rx, Rd = 129, 900, 12, 10
    X = rs.gamma(0.5, 1.0, (F, Rx)) @ rs.gamma(0.3, 1.0, (Rx, T)) + 1e-9
    D = rs.gamma(0.5, 1.0, (F, Rd)) @ rs.gamma(0.3, 1.0, (Rd, T)) + 1e-9
    Y = X + D
    B = rs.random((F, Rx + Rd))
    p = dict(cf="kl", sparsity=5, max_iter=25, conv_eps=1e-3, cost_check=1)
    B1, A1 = run_basis_dnmf(Y, X, D, B, Rx, Rd, p, ctx=gpu_ctx)
    B2, A2 = run_basis_dnmf(Y, X, D, B, Rx, Rd, p, devices=[0, 0])
    assert rel(B2, B1) < 1e-5 and rel(A2, A1) < 1e-5


def test_bad_arguments_are_reported_not_crashed(lib):
    from se_snmf_nat_amd import _lib
    from se_snmf_nat_amd.api import _make_params
    sp = _make_params(64, 10, 8, 1.0, 5, 0.0, 1, True, 0, 0.0, None, None)
    h = C.c_void_p()
    devs = np.zeros(20, np.int32)
    assert lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 20, C.byref(sp), None, C.byref(h)) == 1  # n_dev > 16
    assert lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 11, C.byref(sp), None, C.byref(h)) == 1  # more ranks than frames
    bad = np.asarray([0, 7, 9], np.int64)  # does not end at T
    assert lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 2, C.byref(sp), C.c_void_p(bad.ctypes.data), C.byref(h)) == 1
    devs[0] = 99
    assert lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 1, C.byref(sp), None, C.byref(h)) != 0
    assert b"device" in lib.snmf_last_error()


def test_a_refused_device_list_leaves_no_sticky_error_behind(gpu_ctx, lib):
    """hipGetLastError() is sticky per thread: a device list that is refused (ordinal 99) must not make the NEXT,
    unrelated solve fail at its first kernel-launch check (it did: 'hipGetLastError(): invalid device ordinal')."""
    from se_snmf_nat_amd import sparse_nmf
    from se_snmf_nat_amd.api import _make_params
    sp = _make_params(64, 40, 8, 1.0, 5, 0.0, 1, True, 0, 0.0, None, None)
    h = C.c_void_p()
    devs = np.asarray([0, 99], np.int32)
    assert lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 2, C.byref(sp), None, C.byref(h)) != 0
    V, W0, H0 = synth_problem(64, 40, 8)
    w, _, o = sparse_nmf(V, dict(cf="kl", sparsity=1, max_iter=3, init_w=W0, init_h=H0, cost_check=1), ctx=gpu_ctx)
    assert np.isfinite(w).all() and o["n_iter"] == 3


def test_device_list_calls_reuse_or_rebuild_their_team(gpu_ctx, monkeypatch):
    """A device list's set-up (contexts, peer access, gather buffers) is cached per list between calls (csrc/snmf_multi.h: MultiTeam);
    SNMF_TEAM_CACHE=0 tears it down with the handle.  Either way a call must not see anything of its predecessors: the same
    problem gives the same bits, a LARGER exchange after a smaller one (the gather buffers are re-allocated under the
    cached team) and a different rank count in between included."""
    from se_snmf_nat_amd import sparse_nmf
    small = dict(F=129, T=700, r=24, cf="kl", sparsity=0.3, max_iter=8, conv_eps=0.0, n=2)
    big = dict(F=257, T=2100, r=64, cf="kl", sparsity=5.0, max_iter=8, conv_eps=0.0, n=2)
    out = {}
    for cache in ("2", "0"):
        monkeypatch.setenv("SNMF_TEAM_CACHE", cache)
        res = []
        for case, n in ((small, 2), (big, 2), (small, 3), (big, 2), (small, 2)):
            V, W0, H0 = synth_problem(case["F"], case["T"], case["r"])
            w, h, o = sparse_nmf(V, _p(case, W0, H0), devices=[0] * n)
            res.append((case["F"], n, w, h, o["cost"]))
        out[cache] = res
        assert np.array_equal(res[0][2], res[4][2]) and np.array_equal(res[0][3], res[4][3])   # small, 2 ranks: first and last call
        assert np.array_equal(res[1][2], res[3][2]) and np.array_equal(res[1][3], res[3][3])   # big, 2 ranks
    for a, b in zip(out["2"], out["0"]):
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


def test_a_failed_run_poisons_its_team_and_the_next_handle_starts_fresh(gpu_ctx, lib, monkeypatch):
    """a team went back into the process-wide cache after a FAILED snmf_multi_run, and the next handle on the
    same device list inherited its exchange number, parity and arrival words (in FLAGS mode a peer may have posted arrival
    values ahead of the number the failed rank recorded: the next run would skip waits and sum stale slots).  A rank failure
    is injected (SNMF_MULTI_TEST_FAIL = rank:iteration); the call must fail, the team must NOT be cached, and a second solve
    on the same device list must equal a solve on a process that never failed -- bit for bit.  Then the cache control:
    snmf_multi_release_cache() empties the idle teams."""
    from se_snmf_nat_amd import _lib, release_device_lists
    from se_snmf_nat_amd.api import _make_params
    F, T, r, iters = 129, 900, 24, 7
    V, W0, H0 = synth_problem(F, T, r)
    Vf, Hf, Wf = np.asfortranarray(V, np.float32), np.asfortranarray(H0, np.float32), np.asfortranarray(W0)
    devs = np.asarray([0, 0, 0], np.int32)  # a device list no other test of this module uses (3 ranks on device 0)

    def solve(expect_fail):
        sp = _make_params(F, T, r, 1.0, iters, 0.0, 1, True, 0, 5.0, None, None)
        h = C.c_void_p()
        _lib.check(lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), 3, C.byref(sp), None, C.byref(h)))
        try:
            _lib.check(lib.snmf_multi_set_v_f32(h, C.c_void_p(Vf.ctypes.data), F))
            _lib.check(lib.snmf_multi_set_w_f64(h, C.c_void_p(Wf.ctypes.data), F))
            _lib.check(lib.snmf_multi_set_h_f32(h, C.c_void_p(Hf.ctypes.data), r))
            _lib.check(lib.snmf_multi_init(h))
            done = C.c_int32()
            rc = lib.snmf_multi_run(h, iters, C.byref(done))
            if expect_fail:
                assert rc != 0 and b"injected failure" in lib.snmf_last_error()
                return None
            _lib.check(rc)
            W = np.empty((F, r), order="F")
            Hh = np.empty((r, T), order="F")
            _lib.check(lib.snmf_multi_get_w_f64(h, C.c_void_p(W.ctypes.data), F))
            _lib.check(lib.snmf_multi_get_h_f64(h, C.c_void_p(Hh.ctypes.data), r))
            return W, Hh
        finally:
            lib.snmf_multi_destroy(h)

    release_device_lists()
    assert lib.snmf_multi_cached_teams() == 0
    w_ref, h_ref = solve(False)
    assert lib.snmf_multi_cached_teams() == 1      # a healthy team is kept for the next handle
    monkeypatch.setenv("SNMF_MULTI_TEST_FAIL", "1:3")
    solve(True)
    monkeypatch.delenv("SNMF_MULTI_TEST_FAIL")
    assert lib.snmf_multi_cached_teams() == 0      # ... a poisoned one is destroyed with its last user
    w2, h2 = solve(False)
    assert np.array_equal(w2, w_ref) and np.array_equal(h2, h_ref)
    assert lib.snmf_multi_cached_teams() == 1
    assert release_device_lists() == 1 and lib.snmf_multi_cached_teams() == 0
