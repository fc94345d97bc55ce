"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the fp64 oracle.

Tolerances (north_star: "W/H within 1e-4 relative of reference", fp32 MFMA vs fp64 oracle):
  REL_WH   = 1e-4  Frobenius-relative error on W and on H after identical (V, r, init, iters)
  REL_COST = 1e-5  relative error on every recorded objective value (fp64 accumulation on device)
  ABS_DIV  = 2e-7 * sum(V): each divergence term v*log(v/lam) - v + lam is evaluated in fp32 and
             cancels to O((v-lam)^2/lam), so the absolute error floor of the sum is ~eps_f32 * sum(V)
             (only visible when the fit is near-perfect, div << sum(V))
and the early-stop iteration index must match EXACTLY.
"""
import glob
import os
import re

import numpy as np
import pytest

from oracle.sparse_nmf_oracle import run_basis_dnmf_solves, sparse_nmf as oracle_nmf, synth_problem
from test_oracle import GOLD, SOLVE_CASES, load_case

pytestmark = pytest.mark.gpu
REL_WH = 1e-4
REL_COST = 1e-5


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def check(res, ref, *, cost=True, vsum=0.0):
    w, h, o = res
    wr, hr, orf = ref
    assert o["n_iter"] == orf["n_iter"]
    assert np.isfinite(w).all() and np.isfinite(h).all()
    assert rel(w, wr) < REL_WH, rel(w, wr)
    assert rel(h, hr) < REL_WH, rel(h, hr)
    if cost:
        assert len(o["cost"]) == len(orf["cost"])
        np.testing.assert_allclose(o["cost"], orf["cost"], rtol=REL_COST, atol=2e-7 * vsum)
        np.testing.assert_allclose(o["div"], orf["div"], rtol=REL_COST, atol=2e-7 * vsum)


@pytest.mark.parametrize("path", SOLVE_CASES, ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_golden_vectors_through_c_abi(gpu_ctx, path, dtype):
    from se_snmf_nat_amd import sparse_nmf
    d, p = load_case(path)
    w, h, o = sparse_nmf(d["V"], p, ctx=gpu_ctx, dtype=dtype)
    assert o["n_iter"] == int(d["n_iter"])  # exact stop index
    if "W" in d:
        assert rel(w, d["W"]) < REL_WH
    assert rel(h, d["H"]) < REL_WH
    np.testing.assert_allclose(o["cost"], d["cost"], rtol=REL_COST)
    np.testing.assert_allclose(o["div"], d["div"], rtol=REL_COST)
    np.testing.assert_allclose(np.sqrt((w.astype(np.float64) ** 2).sum(0)), 1.0, rtol=1e-6)


def test_golden_dnmf_three_solve_loop(gpu_ctx):
    """run_basis_DNMF.m:36-55 through the host mirror."""
    from se_snmf_nat_amd import run_basis_dnmf
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    d = dict(np.load(os.path.join(GOLD, "dnmf_loop_513x64_r20_20.npz")))
    Y = ref["Y"]
    X = (Y * d["mask"] + 1e-9).astype(np.float32)
    D = (Y - X + 2e-9).astype(np.float32)
    Bs = np.concatenate([ref["B"][:, :20], ref["B"][:, 100:120]], axis=1)
    p = dict(cf="kl", sparsity=5, max_iter=30, conv_eps=1e-3, cost_check=1)
    B_hat, A_hat = run_basis_dnmf(Y, X, D, Bs, 20, 20, p, ctx=gpu_ctx)
    assert rel(B_hat, d["B_hat"]) < REL_WH
    assert rel(A_hat, d["A_hat"]) < REL_WH


CASES = [
    # name, F, T, r, params, w_ind, h_ind
    ("kl_aligned", 64, 96, 32, dict(cf="kl", sparsity=5, max_iter=20), None, None),
    ("kl_c1_257x2000_r40_50it", 257, 2000, 40, dict(cf="kl", sparsity=5, max_iter=50), None, None),
    ("kl_stop", 257, 2000, 40, dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3), None, None),
    ("kl_honly", 257, 640, 40, dict(cf="kl", sparsity=5, max_iter=20), "none", None),
    ("kl_wonly", 257, 640, 40, dict(cf="kl", sparsity=5, max_iter=20), None, "none"),
    ("kl_wonly_stop", 257, 640, 40, dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3), None, "none"),
    ("kl_honly_stop", 257, 640, 40, dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3), "none", None),
    ("kl_semi", 257, 640, 40, dict(cf="kl", sparsity=5, max_iter=20), "half", None),
    ("kl_neither", 65, 100, 8, dict(cf="kl", sparsity=5, max_iter=6, conv_eps=1e-3), "none", "none"),
    ("kl_513x100_r50", 513, 100, 50, dict(cf="kl", sparsity=5, max_iter=20), None, None),
    ("kl_T1", 513, 1, 200, dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3), "none", None),
    ("kl_F1", 1, 50, 3, dict(cf="kl", sparsity=0.1, max_iter=10), None, None),
    ("kl_r1", 40, 50, 1, dict(cf="kl", sparsity=0.1, max_iter=10), None, None),
    ("kl_ragged", 37, 131, 13, dict(cf="kl", sparsity=0.5, max_iter=15), None, None),
    ("kl_r256", 257, 4096, 256, dict(cf="kl", sparsity=5, max_iter=5), None, None),
    ("kl_r512", 130, 700, 512, dict(cf="kl", sparsity=1, max_iter=4), None, None),
    ("kl_r600_honly", 100, 200, 600, dict(cf="kl", sparsity=5, max_iter=5), "none", None),
    ("kl_r600_full", 97, 200, 600, dict(cf="kl", sparsity=5, max_iter=5), None, None),
    ("ed_r700_full", 65, 150, 700, dict(cf="ed", sparsity=1, max_iter=4), None, None),
    ("kl_F1025", 1025, 300, 20, dict(cf="kl", sparsity=5, max_iter=5), None, None),
    ("kl_nocheck", 257, 640, 40, dict(cf="kl", sparsity=5, max_iter=20, cost_check=0), None, None),
    ("kl_maxiter1", 65, 100, 8, dict(cf="kl", sparsity=5, max_iter=1, conv_eps=1e-3), None, None),
    ("kl_maxiter2", 65, 100, 8, dict(cf="kl", sparsity=5, max_iter=2, conv_eps=1e-3), None, None),
    ("kl_lambda0", 65, 100, 8, dict(cf="kl", sparsity=0, max_iter=10), None, None),
    ("ed_full", 257, 640, 40, dict(cf="ed", sparsity=5, max_iter=20), None, None),
    ("ed_wonly", 257, 640, 40, dict(cf="ed", sparsity=5, max_iter=20), None, "none"),
    ("ed_honly", 257, 640, 40, dict(cf="ed", sparsity=5, max_iter=20), "none", None),
    ("ed_c5_scaled_513x2048_r512", 513, 2048, 512, dict(cf="ed", sparsity=50, max_iter=4), None, None),
    ("is_full", 257, 640, 40, dict(cf="is", sparsity=0.1, max_iter=20), None, None),
    ("b05_full", 257, 640, 40, dict(cf="beta", beta=0.5, sparsity=1, max_iter=20), None, None),
    ("b15_r300", 129, 300, 300, dict(cf="beta", beta=1.5, sparsity=1, max_iter=10), None, None),
    ("b3_full", 65, 200, 16, dict(cf="beta", beta=3.0, sparsity=1, max_iter=10), None, None),
    # F + r > 1272: the 32-frame LDS images do not fit -> 16-frame tiles (the reference's exemplar setting R_x = R_d = 500
    # at F = 513, settings/bak_IS16_results/initial_setting_Exemplar.m:47-48; src/bnmf_sep_event_RT_IS16.m:138-154)
    # small rank / few rows with MANY tiles per workgroup (T > 256 x 32): role-pipeline waves that own no tile must not
    # run ahead of the working ones (the LDS counters are totals)
    ("kl_r8_T20000", 257, 20000, 8, dict(cf="kl", sparsity=5, max_iter=4), None, None),
    ("kl_r70_T12000_F65", 65, 12000, 70, dict(cf="kl", sparsity=1, max_iter=4), None, None),
    ("kl_r200_T9000_F161", 161, 9000, 200, dict(cf="kl", sparsity=1, max_iter=3), None, None),
    ("kl_r96_T20000_F97_honly", 97, 20000, 96, dict(cf="kl", sparsity=1, max_iter=3), "none", None),
    ("kl_r1000_honly", 513, 64, 1000, dict(cf="kl", sparsity=5, max_iter=8, conv_eps=1e-3), "none", None),
    ("kl_r1000_full", 513, 300, 1000, dict(cf="kl", sparsity=5, max_iter=4), None, None),
    ("kl_r1000_wonly", 513, 200, 1000, dict(cf="kl", sparsity=5, max_iter=4), None, "none"),
    ("ed_r1000_full", 513, 200, 1000, dict(cf="ed", sparsity=1, max_iter=3), None, None),
    ("b05_r800_513", 513, 150, 800, dict(cf="beta", beta=0.5, sparsity=1, max_iter=3), None, None),
    # the reference's shipped geometry (513 rows, R = 100 / 200) with more than one tile per workgroup (T > 256 x 32), so
    # that the half-tile role pipeline k_hstep_rh, k_wstats on eight consumer waves with the leftover columns on the VALU,
    # its objective-carrying W-only launch and the trimmed contraction depth (13 / 25 k-blocks) are what runs
    ("kl_f513_r100_T8400_full", 513, 8400, 100, dict(cf="kl", sparsity=5, max_iter=6), None, None),
    ("kl_f513_r200_T8400_honly_stop", 513, 8400, 200, dict(cf="kl", sparsity=5, max_iter=60, conv_eps=2e-3), "none", None),
    ("kl_f513_r100_T8400_wonly", 513, 8400, 100, dict(cf="kl", sparsity=5, max_iter=6), None, "none"),
    ("kl_f513_r104_T8400_full", 513, 8400, 104, dict(cf="kl", sparsity=5, max_iter=4), None, None),  # 8 leftover columns
    ("kl_f385_r36_T9000_full", 385, 9000, 36, dict(cf="kl", sparsity=1, max_iter=4), None, None),   # 12 row tiles, 4 leftover columns
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_synthetic_cases_against_oracle(gpu_ctx, case):
    from se_snmf_nat_amd import sparse_nmf
    name, F, T, r, params, wi, hi = case
    V, W0, H0 = synth_problem(F, T, r)
    p = dict(params, init_w=W0, init_h=H0)
    p.setdefault("cost_check", 1)
    if wi == "none":
        p["w_update_ind"] = np.zeros(r, bool)
    elif wi == "half":
        p["w_update_ind"] = np.arange(r) >= r // 2
    if hi == "none":
        p["h_update_ind"] = np.zeros(r, bool)
    check(sparse_nmf(V, p, ctx=gpu_ctx), oracle_nmf(V, p), vsum=float(V.sum()))


def test_power_domain_dynamic_range(gpu_ctx):
    """|STFT|^2 of int16 audio reaches ~1e10: fp32 range/precision check."""
    from se_snmf_nat_amd import sparse_nmf
    V, W0, H0 = synth_problem(257, 2000, 40, scale="power")
    p = dict(cf="kl", sparsity=5, max_iter=30, init_w=W0, init_h=H0, cost_check=1)
    check(sparse_nmf(V, p, ctx=gpu_ctx), oracle_nmf(V, p))


def test_sparsity_argument_forms(gpu_ctx):
    """scalar / r x 1 column / full r x n forms of p.sparsity (src/sparse_nmf.m:150-155)."""
    from se_snmf_nat_amd import sparse_nmf
    V, W0, H0 = synth_problem(129, 300, 24)
    rs = np.random.RandomState(3)
    for sp in (np.linspace(0, 9, 24).reshape(-1, 1), np.abs(rs.randn(24, 300)) * 4, np.full((24, 300), 2.0)):
        for cf in ("kl", "ed"):
            p = dict(cf=cf, sparsity=sp, max_iter=15, init_w=W0, init_h=H0, cost_check=1)
            check(sparse_nmf(V, p, ctx=gpu_ctx), oracle_nmf(V, p))


def test_sparsity_forms_on_the_pipelined_kernels(gpu_ctx):
    """r x 1 and full r x n sparsity (src/sparse_nmf.m:150-155) through the pipelined KL kernels: a shape whose last
    partial round is split 4 ways (282 tiles on 256 workgroups: the last-arriver finishing pass forms dph from S) and the
    reference's F = 513 at R = 100 (k_hstep_rh)."""
    from se_snmf_nat_amd import sparse_nmf
    for F, T, r in ((257, 9000, 40), (513, 8400, 100)):
        V, W0, H0 = synth_problem(F, T, r)
        rs = np.random.RandomState(F)
        for sp in (np.linspace(0, 9, r).reshape(-1, 1), np.abs(rs.randn(r, T)) * 4):
            p = dict(cf="kl", sparsity=sp, max_iter=4, init_w=W0, init_h=H0, cost_check=1)
            check(sparse_nmf(V, p, ctx=gpu_ctx), oracle_nmf(V, p))


def test_gpu_variant_entry_point(gpu_ctx):
    """sparse_nmf_GPU.m deltas: no V floor, objective vectors left zero, cost_check ignored."""
    from se_snmf_nat_amd import sparse_nmf_GPU
    V, W0, H0 = synth_problem(65, 200, 12)
    p = dict(cf="kl", sparsity=1, max_iter=200, conv_eps=1e-3, init_w=W0, init_h=H0)
    w, h, o = sparse_nmf_GPU(V, p, ctx=gpu_ctx)
    wr, hr, orf = oracle_nmf(V, p, gpu_variant=True)
    assert o["n_iter"] == orf["n_iter"] < 200
    assert len(o["div"]) == 200 and not o["div"].any() and not o["cost"].any()
    assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH


def test_display_prints_the_reference_lines(gpu_ctx, capsys):
    """p.display (src/sparse_nmf.m:181-183,266-270,276-278,288-290): the drop-in writes what the reference writes -- here
    compared with the oracle's console output on the same problem: every character outside the %.3e numbers equal (so
    also the count of backspaces, i.e. the printed widths), the numbers within one unit of their last printed digit
    (full update with early stop, no cost_check, the default display = 0, the _GPU file's per-iteration lines, and
    init_h = 'ones' with its unconditional notice :135)."""
    import re
    from se_snmf_nat_amd import sparse_nmf, sparse_nmf_GPU
    num = re.compile(r"\d\.\d{3}e[+-]\d{2}")
    V, W0, H0 = synth_problem(65, 200, 12)
    base = dict(cf="kl", sparsity=1, max_iter=200, conv_eps=1e-3, init_w=W0, init_h=H0, cost_check=1)
    for extra, gpu in ((dict(display=1), False), (dict(display=1, cost_check=0, max_iter=4), False), (dict(), False),
                       (dict(display=2, conv_eps=0, max_iter=5), False), (dict(display=1, max_iter=6, conv_eps=0), True),
                       (dict(display=1, init_h="ones", max_iter=3), False)):
        p = dict(base, **extra)
        (sparse_nmf_GPU if gpu else sparse_nmf)(V, p, ctx=gpu_ctx)
        got = capsys.readouterr().out
        oracle_nmf(V, p, gpu_variant=gpu)
        want = capsys.readouterr().out
        if isinstance(p["init_h"], str):
            assert got.startswith("sup_nmf: Initalizing H with ones.\n")
            got = got[len("sup_nmf: Initalizing H with ones.\n"):]
        assert num.sub("#", got) == num.sub("#", want), (extra, got[:200], want[:200])
        for a, b in zip(num.findall(got), num.findall(want)):
            assert abs(float(a) - float(b)) <= 1.01e-3 * abs(float(b)), (a, b)
        if p.get("display", 0) == 0:
            assert got == ""
        elif not gpu:
            assert got.endswith("\\nMax Iteration reached, aborting iteration\\n\n")


def test_inputs_are_not_mutated_and_random_init_path(gpu_ctx):
    from se_snmf_nat_amd import sparse_nmf
    V, W0, H0 = synth_problem(40, 60, 6)
    Vc, Wc, Hc = V.copy(), W0.copy(), H0.copy()
    sparse_nmf(V, dict(init_w=W0, init_h=H0, max_iter=3, cost_check=1), ctx=gpu_ctx)
    assert np.array_equal(V, Vc) and np.array_equal(W0, Wc) and np.array_equal(H0, Hc)
    # p.r only: random factors from the seeded stand-in generator; r > size(init_w,2): appended columns
    w, h, o = sparse_nmf(V, dict(r=5, max_iter=30, cost_check=1, random_seed=7), ctx=gpu_ctx)
    assert w.shape == (40, 5) and h.shape == (5, 60) and np.all(np.diff(o["cost"]) <= 0)
    w2, h2, _ = sparse_nmf(V, dict(r=5, max_iter=30, cost_check=1, random_seed=7), ctx=gpu_ctx)
    assert np.array_equal(w, w2) and np.array_equal(h, h2)  # deterministic
    w, h, _ = sparse_nmf(V, dict(init_w=W0, r=9, init_h="ones", max_iter=3, cost_check=1), ctx=gpu_ctx)
    assert w.shape == (40, 9)


def test_partial_h_update_ind_is_a_dimension_error(gpu_ctx):
    from se_snmf_nat_amd import SnmfError, sparse_nmf
    V, W0, H0 = synth_problem(40, 60, 6)
    with pytest.raises(SnmfError, match="DIM"):
        sparse_nmf(V, dict(init_w=W0, init_h=H0, h_update_ind=np.array([1, 1, 0, 1, 1, 1], bool), cost_check=1),
                   ctx=gpu_ctx)


def test_runs_are_bitwise_reproducible(gpu_ctx):
    """Fixed-order partial sums everywhere: two runs give identical bits (needed so that W replicas
    stay identical across ranks)."""
    from se_snmf_nat_amd import sparse_nmf
    V, W0, H0 = synth_problem(257, 3000, 64)
    p = dict(cf="kl", sparsity=5, max_iter=10, init_w=W0, init_h=H0, cost_check=1)
    a = sparse_nmf(V, p, ctx=gpu_ctx)
    b = sparse_nmf(V, p, ctx=gpu_ctx)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2]["cost"], b[2]["cost"])


def test_full_size_c2_properties(gpu_ctx):
    """BASELINE C2 (257 x 100000, r = 256, KL): too big for the oracle in seconds, so check
    size-independent properties: non-increasing cost, unit-norm non-negative W, finite H, and frame
    locality: the H-only solve of a column block equals the same columns of the full H-only solve."""
    from se_snmf_nat_amd import Plan
    F, T, r = 257, 100_000, 256
    rs = np.random.default_rng(0)
    Wt = rs.gamma(0.5, 1.0, (F, r)).astype(np.float32)
    Ht = rs.gamma(0.3, 1.0, (r, T)).astype(np.float32)
    V = Wt @ Ht + 1e-9
    W0 = rs.random((F, r)).astype(np.float32)
    H0 = rs.random((r, T)).astype(np.float32)
    plan = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=12, conv_eps=0.0, cost_check=True, sparsity=5.0)
    plan.set_v(V); plan.set_w(W0); plan.set_h(H0); plan.init()
    assert plan.run() == 12
    w, h = plan.get_w(), plan.get_h()
    div, cost, n = plan.get_objective()
    assert n == 12 and np.all(np.diff(cost) <= 0) and np.all(cost > 0)
    np.testing.assert_allclose(np.sqrt((w ** 2).sum(0)), 1.0, rtol=1e-6)
    assert (w >= 0).all() and (h >= 0).all() and np.isfinite(h).all()
    plan.close()
    # frame locality of the H-only solve (the property that makes frame sharding exact)
    hp = dict(beta=1.0, max_iter=5, conv_eps=0.0, cost_check=True, sparsity=5.0, w_update_ind=np.zeros(r, bool))
    full = Plan(gpu_ctx, F, T, r, **hp)
    full.set_v(V); full.set_w(W0); full.set_h(H0); full.init(); full.run()
    hf = full.get_h(np.float32)
    full.close()
    t0, t1 = 33_333, 41_111
    part = Plan(gpu_ctx, F, t1 - t0, r, **hp)
    part.set_v(V[:, t0:t1]); part.set_w(W0); part.set_h(H0[:, t0:t1]); part.init(); part.run()
    hpart = part.get_h(np.float32)
    part.close()
    assert rel(hpart, hf[:, t0:t1]) < 1e-6


def test_full_size_c2_against_the_oracle_golden(gpu_ctx):
    """BASELINE C2 at FULL size (257 x 100000, r = 256, KL, sparsity 5) against the fp64 oracle: the committed golden
    tests/golden/c2_full_257x100000_r256.npz (tests/golden/make_golden_c2.py, ~12 min of oracle time, run once) holds
    W and the head / tail frames of H after 12 iterations and the cost of every iterate on exactly the inputs bench.py
    times (bench.make_problem, V and H0 rounded to fp32).  Tolerances: W, H within 1e-4 relative (north_star), every
    cost within 1e-6 relative (src/sparse_nmf.m:248-264)."""
    from bench import F_, R_, SPARSITY, T_, make_problem
    from se_snmf_nat_amd import Plan
    g = np.load(os.path.join(GOLD, "c2_full_257x100000_r256.npz"))
    assert (int(g["F"]), int(g["T"]), int(g["r"])) == (F_, T_, R_)
    V, W0, H0 = make_problem(F_, T_, R_)
    plan = Plan(gpu_ctx, F_, T_, R_, beta=1.0, max_iter=12, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
    plan.set_v(V.astype(np.float32)); plan.set_w(W0); plan.set_h(H0.astype(np.float32)); plan.init()
    assert plan.run() == 12
    w, h = plan.get_w(), plan.get_h()
    div, cost, n = plan.get_objective()
    plan.close()
    assert n == 12
    np.testing.assert_allclose(cost[:12], g["cost"][:12], rtol=1e-6)
    np.testing.assert_allclose(div[:12], g["div"][:12], rtol=1e-6)
    ew, eh0, eh1 = rel(w, g["W12"]), rel(h[:, :64], g["H12_head"]), rel(h[:, -64:], g["H12_tail"])
    print(f"C2 full size vs oracle: relW={ew:.2e} relH(head)={eh0:.2e} relH(tail)={eh1:.2e} "
          f"relcost={abs(cost[11] - g['cost'][11]) / g['cost'][11]:.2e}")
    assert ew < REL_WH and eh0 < REL_WH and eh1 < REL_WH


def test_plan_step_api_equals_run(gpu_ctx):
    """hstep / wstats / wapply with a device statistics buffer (the multi-GPU path at world size 1,
    driven by se_snmf_nat_amd.dist.ShardedTrainer over torch) gives the same bits as plan.run."""
    torch = pytest.importorskip("torch")
    from se_snmf_nat_amd import sparse_nmf
    from se_snmf_nat_amd.dist import ShardedTrainer
    V, W0, H0 = synth_problem(257, 1000, 48)
    for conv_eps, kw in ((0.0, {}), (1e-3, {}), (1e-3, dict(w_update_ind=np.zeros(48, bool)))):
        p = dict(cf="kl", sparsity=5, max_iter=60, conv_eps=conv_eps, init_w=W0, init_h=H0, cost_check=1, **kw)
        w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
        tr = ShardedTrainer(V, W0, H0, beta=1.0, sparsity=5.0, max_iter=60, conv_eps=conv_eps, cost_check=True,
                            device=0, **kw)
        tr.run()
        tr.sync()
        w2, h2, (div, cost, n) = tr.result()
        assert n == o["n_iter"]
        assert np.array_equal(w, w2) and np.array_equal(h, h2)
        np.testing.assert_array_equal(cost[:n], o["cost"][:n])


def test_device_resident_inputs(gpu_ctx):
    """V/W/H handed over as device pointers (torch CUDA tensors, column-major)."""
    torch = pytest.importorskip("torch")
    from se_snmf_nat_amd import Plan, sparse_nmf
    V, W0, H0 = synth_problem(129, 500, 24)
    p = dict(cf="kl", sparsity=2, max_iter=10, init_w=W0, init_h=H0, cost_check=1)
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx, dtype=np.float32)
    dev = torch.device("cuda", 0)
    tV = torch.tensor(np.ascontiguousarray(V.T), dtype=torch.float32, device=dev)   # (T, F) = column-major F x T
    tW = torch.tensor(np.ascontiguousarray(W0.T), dtype=torch.float64, device=dev)
    tH = torch.tensor(np.ascontiguousarray(H0.T), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    plan = Plan(gpu_ctx, 129, 500, 24, beta=1.0, max_iter=10, sparsity=2.0)
    plan.set_v(tV); plan.set_w(tW); plan.set_h(tH); plan.init(); plan.run()
    assert rel(plan.get_w(), w) < 1e-6 and rel(plan.get_h(), h) < 1e-6


def test_persistent_small_solve_matches_generic_path_and_oracle(gpu_ctx, monkeypatch):
    """T <= 32 H-only solves (the online call, src/bnmf_sep_event_RT_IS16.m:138-154) run as ONE
    persistent launch; it must agree with the per-iteration path and with the oracle, including the
    stop index, for every divergence."""
    from se_snmf_nat_amd import sparse_nmf
    for (F, T, r, cf, beta) in [(513, 1, 200, "kl", 1), (257, 7, 40, "kl", 1), (129, 32, 24, "ed", 2),
                                (65, 5, 8, "is", 0), (129, 20, 33, "beta", 0.5)]:
        V, W0, H0 = synth_problem(F, T, r)
        p = dict(cf=cf, beta=beta, sparsity=0.5, max_iter=60, conv_eps=1e-3, init_w=W0, init_h=H0, cost_check=1,
                 w_update_ind=np.zeros(r, bool))
        ref = oracle_nmf(V, p)
        monkeypatch.delenv("SNMF_NO_SMALL", raising=False)
        small = sparse_nmf(V, p, ctx=gpu_ctx)
        monkeypatch.setenv("SNMF_NO_SMALL", "1")  # no persistent kernel: the per-iteration plan loop
        generic = sparse_nmf(V, p, ctx=gpu_ctx)
        monkeypatch.setenv("SNMF_NO_SMALL", "2")  # no register-resident frame kernel: k_hsolve_small (MFMA, H in LDS) also at T = 1
        lds_small = sparse_nmf(V, p, ctx=gpu_ctx)
        monkeypatch.delenv("SNMF_NO_SMALL")
        check(small, ref, vsum=float(V.sum()))
        check(generic, ref, vsum=float(V.sum()))
        check(lds_small, ref, vsum=float(V.sum()))
        assert small[2]["n_iter"] == generic[2]["n_iter"] == lds_small[2]["n_iter"]
        assert rel(small[1], generic[1]) < 1e-5
        # no cost_check: fixed iteration count, zero objective vectors
        p2 = dict(p, cost_check=0, max_iter=9)
        a = sparse_nmf(V, p2, ctx=gpu_ctx)
        b = oracle_nmf(V, p2)
        assert a[2]["n_iter"] == 9 and not a[2]["cost"].any() and rel(a[1], b[1]) < REL_WH


def test_online_stream_reuses_the_resident_dictionary(gpu_ctx):
    """Frame loop of the online path: W is set and normalised once, each frame only sets V and H0.
    Every frame must equal an independent sparse_nmf call (the reference re-seeds and re-normalises
    per call, src/sparse_nmf.m:112-114,:157-160)."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    B, Y = ref["B"].astype(np.float64), ref["Y"].astype(np.float64)
    H0 = np.random.RandomState(1).random_sample((200, 1))
    plan = Plan(gpu_ctx, 513, 1, 200, beta=1.0, max_iter=100, conv_eps=1e-3, cost_check=True, sparsity=5.0,
                w_update_ind=np.zeros(200, bool))
    plan.set_w(B)
    for col in (0, 17, 40, 3):
        plan.set_v(Y[:, col:col + 1])
        plan.set_h(H0)
        plan.init()
        n = plan.run()
        h = plan.get_h()
        w1, h1, o1 = sparse_nmf(Y[:, col:col + 1], dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, init_w=B,
                                                         init_h=H0, cost_check=1, w_update_ind=np.zeros(200, bool)),
                                ctx=gpu_ctx)
        assert n == o1["n_iter"]
        assert np.array_equal(h, h1)
        div, cost, nn = plan.get_objective()
        np.testing.assert_array_equal(cost[:nn], o1["cost"])
    # the stream entry point: all frames in one call, one persistent workgroup per frame
    stream = Plan(gpu_ctx, 513, 64, 200, beta=1.0, max_iter=100, conv_eps=1e-3, cost_check=True, sparsity=5.0,
                  w_update_ind=np.zeros(200, bool))
    stream.set_w(B)
    cols = [0, 17, 40, 3, 63]
    for dtype in (np.float64, np.float32):
        Hs, nit, lc = stream.solve_frames(Y[:, cols], H0, dtype=dtype)
        for j, col in enumerate(cols):
            w1, h1, o1 = sparse_nmf(Y[:, col:col + 1], dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, init_w=B,
                                                             init_h=H0, cost_check=1, w_update_ind=np.zeros(200, bool)),
                                    ctx=gpu_ctx, dtype=dtype)
            assert nit[j] == o1["n_iter"] and np.array_equal(Hs[:, j:j + 1], h1) and lc[j] == o1["cost"][-1]
    # all 64 frames against the oracle (stop index exact, H within tolerance)
    Hs, nit, lc = stream.solve_frames(Y, H0)
    for col in range(0, 64, 7):
        wr, hr, orf = oracle_nmf(Y[:, col:col + 1], dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, init_w=B,
                                                          init_h=H0, cost_check=1, w_update_ind=np.zeros(200, bool)))
        assert nit[col] == orf["n_iter"] and rel(Hs[:, col:col + 1], hr) < REL_WH
        assert abs(lc[col] - orf["cost"][-1]) <= REL_COST * orf["cost"][-1]
    # two frames per solve (joint cost over the pair, like a 513 x 2 call)
    H02 = np.random.RandomState(2).random_sample((200, 2))
    Hs2, nit2, _ = stream.solve_frames(Y[:, :6], H02)
    for j in range(3):
        w1, h1, o1 = sparse_nmf(Y[:, 2 * j:2 * j + 2], dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, init_w=B,
                                                             init_h=H02, cost_check=1, w_update_ind=np.zeros(200, bool)),
                                ctx=gpu_ctx)
        assert nit2[j] == o1["n_iter"] and np.array_equal(Hs2[:, 2 * j:2 * j + 2], h1)


def test_sharded_dnmf_loop_world1_equals_host_mirror(gpu_ctx):
    """run_basis_dnmf_sharded (the multi-GPU form of run_basis_DNMF.m:36-55) at world size 1 must give the
    same bits as the unsharded host mirror when fed the same initial activations."""
    pytest.importorskip("torch")
    from se_snmf_nat_amd import run_basis_dnmf
    from se_snmf_nat_amd.dist import run_basis_dnmf_sharded
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    d = dict(np.load(os.path.join(GOLD, "dnmf_loop_513x64_r20_20.npz")))
    Y = ref["Y"].astype(np.float64)
    X = (ref["Y"] * d["mask"] + 1e-9).astype(np.float32).astype(np.float64)
    D = (ref["Y"] - X.astype(np.float32) + 2e-9).astype(np.float32).astype(np.float64)
    Bs = np.concatenate([ref["B"][:, :20], ref["B"][:, 100:120]], axis=1).astype(np.float64)
    p = dict(cf="kl", sparsity=5, max_iter=30, conv_eps=1e-3, cost_check=1, random_seed=1)
    B1, A1 = run_basis_dnmf(Y, X, D, Bs, 20, 20, p, ctx=gpu_ctx)
    B2, A2 = run_basis_dnmf_sharded(Y, X, D, Bs, 20, 20, p, device=0)
    assert np.array_equal(B1, B2) and np.array_equal(A1, A2)


def test_full_size_c2_scale_equivariance(gpu_ctx):
    """KL updates are equivariant under V -> c*V, H0 -> c*H0 (src/sparse_nmf.m:190-222: V./Lam, colsum(W)+S and the
    W ratio dmw./dpw do not see c), so W must not move and H, cost must scale by c.  With c = 4 every fp32
    operation scales exactly (no floor is active on this data), which makes it a checksum of the whole
    257 x 100000, r = 256 pipeline that needs no oracle."""
    from se_snmf_nat_amd import Plan
    F, T, r = 257, 100_000, 256
    rs = np.random.default_rng(1)
    V = (rs.gamma(0.5, 1.0, (F, r)).astype(np.float32) @ rs.gamma(0.3, 1.0, (r, T)).astype(np.float32) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r)).astype(np.float32)
    H0 = (rs.random((r, T)).astype(np.float32) + 0.01).astype(np.float32)
    out = []
    for c in (1.0, 4.0):
        plan = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=6, conv_eps=0.0, cost_check=True, sparsity=5.0)
        plan.set_v(V * np.float32(c)); plan.set_w(W0); plan.set_h(H0 * np.float32(c)); plan.init(); plan.run()
        out.append((plan.get_w(np.float32), plan.get_h(np.float32), plan.get_objective()[1].copy()))
        plan.close()
    (w1, h1, c1), (w4, h4, c4) = out
    assert rel(w4, w1) < 1e-6 and rel(h4, 4 * h1) < 1e-6
    np.testing.assert_allclose(c4, 4 * c1, rtol=1e-6)


def test_full_size_c5_properties(gpu_ctx):
    """BASELINE C5 (513 x 500000, r = 512, beta = 2, lambda = 50): decreasing cost (the reference's beta = 2 update
    is not monotone from a random start -- the fp64 oracle shows the same +0.1 % wobble at iteration 3 on a
    513 x 3000 analogue -- so only first vs last is asserted), unit-norm non-negative W, finite H, and frame
    locality of the H-only solve at full size."""
    from se_snmf_nat_amd import Plan
    F, T, r = 513, 500_000, 512
    rs = np.random.default_rng(2)
    V = rs.gamma(0.5, 1.0, (F, 64)).astype(np.float32) @ rs.gamma(0.3, 1.0, (64, T)).astype(np.float32) + np.float32(1e-3)
    W0 = rs.random((F, r), dtype=np.float32)
    H0 = rs.random((r, T), dtype=np.float32)
    plan = Plan(gpu_ctx, F, T, r, beta=2.0, max_iter=4, conv_eps=0.0, cost_check=True, sparsity=50.0)
    plan.set_v(V); plan.set_w(W0); plan.set_h(H0); plan.init()
    assert plan.run() == 4
    w = plan.get_w(np.float32)
    div, cost, n = plan.get_objective()
    assert n == 4 and cost[-1] < cost[0] and np.all(cost > 0)
    np.testing.assert_allclose(np.sqrt((w.astype(np.float64) ** 2).sum(0)), 1.0, rtol=1e-6)
    h = plan.get_h(np.float32)
    assert (w >= 0).all() and (h >= 0).all() and np.isfinite(h).all()
    plan.close()
    del h
    hp = dict(beta=2.0, max_iter=2, conv_eps=0.0, cost_check=True, sparsity=50.0, w_update_ind=np.zeros(r, bool))
    full = Plan(gpu_ctx, F, T, r, **hp)
    full.set_v(V); full.set_w(W0); full.set_h(H0); full.init(); full.run()
    t0, t1 = 123_457, 131_071
    hf = full.get_h(np.float32)[:, t0:t1].copy()
    full.close()
    part = Plan(gpu_ctx, F, t1 - t0, r, **hp)
    part.set_v(np.ascontiguousarray(V[:, t0:t1])); part.set_w(W0); part.set_h(np.ascontiguousarray(H0[:, t0:t1])); part.init(); part.run()
    assert rel(part.get_h(np.float32), hf) < 1e-6
    part.close()


def test_full_size_c4_dnmf_properties(gpu_ctx):
    """BASELINE C4 shape on one GPU (F = 513, T = 100000, R_x = R_d = 100): the 3-solve loop of run_basis_DNMF.m:36-55.
    Size-independent properties: B_hat has unit-norm non-negative columns, A_hat is finite and non-negative, and the
    result equals the frame-sharded implementation at world size 1 bit for bit."""
    from se_snmf_nat_amd import run_basis_dnmf
    from se_snmf_nat_amd.dist import run_basis_dnmf_sharded
    F, T, R = 513, 100_000, 100
    rs = np.random.default_rng(3)
    X = rs.gamma(0.5, 1.0, (F, 40)).astype(np.float32) @ rs.gamma(0.3, 1.0, (40, T)).astype(np.float32) + np.float32(1e-9)
    D = rs.gamma(0.5, 1.0, (F, 40)).astype(np.float32) @ rs.gamma(0.3, 1.0, (40, T)).astype(np.float32) + np.float32(1e-9)
    Y = X + D
    B = rs.random((F, 2 * R)).astype(np.float64)
    p = dict(cf="kl", sparsity=5, max_iter=4, conv_eps=0.0, cost_check=1, random_seed=1)
    B1, A1 = run_basis_dnmf(Y, X, D, B, R, R, p, ctx=gpu_ctx, dtype=np.float32)
    np.testing.assert_allclose(np.sqrt((B1.astype(np.float64) ** 2).sum(0)), 1.0, rtol=1e-6)
    assert (B1 >= 0).all() and (A1 >= 0).all() and np.isfinite(A1).all()
    B2, A2 = run_basis_dnmf_sharded(Y, X, D, B, R, R, p, device=0)
    assert rel(B2, B1) < 1e-6 and rel(A2, A1) < 1e-6


def test_dnmf_adapt_caller(gpu_ctx):
    """src/DNMF_adapt.m:1-20 (H-only on Y, then W-only on D for the noise columns) against the same two oracle calls."""
    from se_snmf_nat_amd import dnmf_adapt
    V, W0, _ = synth_problem(129, 300, 14)
    D, _, _ = synth_problem(129, 300, 14, seed_data=5)
    p = dict(cf="kl", sparsity=1.0, max_iter=25, conv_eps=1e-3, cost_check=1, random_seed=1, R_x=8, R_d=6)
    B_a = dnmf_adapt(V, D, W0, p, ctx=gpu_ctx)
    q = dict(p, w_update_ind=np.zeros(14, bool), h_update_ind=np.ones(14, bool), init_w=W0)
    _, A, _ = oracle_nmf(V, q)
    q = dict(p, w_update_ind=np.ones(6, bool), h_update_ind=np.zeros(6, bool), init_w=W0[:, 8:], init_h=A[8:, :])
    ref, _, _ = oracle_nmf(D, q)
    assert B_a.shape == (129, 6) and rel(B_a, ref) < REL_WH


@pytest.mark.parametrize("F,r,T,mode", [(513, 100, 8300, "full"), (513, 97, 8200, "h"), (512, 98, 8500, "full"), (513, 99, 16500, "h"),
                                        (512, 100, 8193, "semi"),
                                        # r = 193..200: the B waves in pairs (R_x + R_d = 200 at F = 513, run_basis_DNMF.m:40); 196 / 193:
                                        # the second leftover group is all / mostly padding; 16500 frames: with a split last round
                                        (513, 200, 8300, "h"), (513, 200, 8200, "full"), (512, 196, 8500, "h"), (513, 193, 16500, "h"),
                                        (513, 197, 8193, "semi")])
def test_h_step_cut_over_the_contraction(gpu_ctx, F, r, T, mode):
    """r = 97..100 and r = 193..200 on 16 row tiles (the reference's R = 100 and R_x + R_d = 200 at F = 513,
    settings/initial_setting_SNMF_NAT.m:48-49): k_hstep_rh cuts P2 over the contraction (four ways / in wave pairs) and takes
    the leftover columns as 4x4x1 MFMAs on the same ratio fragments (src/sparse_nmf.m:189-195 is what it computes)."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    rs = np.random.default_rng(F + r + T)
    V = rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3
    p = dict(cf="kl", sparsity=rs.random(r) * 4, max_iter=4, conv_eps=0, cost_check=1, init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    kw = {}
    if mode == "h":
        p["w_update_ind"] = kw["w_update_ind"] = np.zeros(r, bool)
    if mode == "semi":
        p["w_update_ind"] = kw["w_update_ind"] = np.arange(r) >= 50
    pl = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=2, cost_check=True, **kw)
    assert "over the contraction" in pl.describe()
    pl.close()
    check(sparse_nmf(V, p, ctx=gpu_ctx), oracle_nmf(V, p))


@pytest.mark.parametrize("F,r,T,mode,spk", [(64, 100, 9000, "full", "scalar"), (64, 100, 8210, "h", "rvec"), (64, 100, 8500, "full", "matrix"),
                                            (40, 100, 12000, "full", "rvec"), (64, 128, 9000, "semi", "scalar"), (32, 20, 9000, "full", "matrix"),
                                            (64, 40, 300, "full", "rvec"), (64, 100, 8600, "w", "scalar"), (48, 70, 33, "h", "matrix"),
                                            # R_x + R_d = 200 on Mel bands: solve 1 of run_basis_DNMF_Mel.m:75
                                            (64, 200, 9000, "h", "scalar"), (64, 256, 8300, "full", "rvec"), (64, 161, 8400, "h", "matrix")])
def test_small_f_kernels_against_oracle(gpu_ctx, F, r, T, mode, spk):
    """The Mel solves (run_basis_train.m:90-91: 64 bands, R = 100; run_basis_DNMF_Mel.m:75-88): F <= 64 and r <= 256 take
    k_hstep_sf for the KL H update (a tile per wave, operands loaded straight into MFMA layout, csrc/snmf_smallf.h) and k_wstats with
    consumer teams; every sparsity form (src/sparse_nmf.m:150-155), every update mode, edge tiles in F and in T."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    rs = np.random.default_rng(F * 7 + r + T)
    V = rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3
    sp = {"scalar": 3.0, "rvec": rs.random(r) * 4, "matrix": np.abs(rs.standard_normal((r, T))) * 3}[spk]
    p = dict(cf="kl", sparsity=sp, max_iter=5, conv_eps=0, cost_check=1, init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    kw = {}
    if mode == "h":
        p["w_update_ind"] = kw["w_update_ind"] = np.zeros(r, bool)
    if mode == "w":
        p["h_update_ind"] = kw["h_update_ind"] = np.zeros(r, bool)
    if mode == "semi":
        p["w_update_ind"] = kw["w_update_ind"] = np.arange(r) >= r // 2
    pl = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=2, cost_check=True, **kw)
    assert ("k_hstep_sf" in pl.describe()) == (mode != "w")
    pl.close()
    check(sparse_nmf(V, p, ctx=gpu_ctx), oracle_nmf(V, p))


def test_euclidean_w_step_through_the_gram_matrix(gpu_ctx):
    """beta = 2, r > 256, full update: P = max(W*H, flr) * H' (src/sparse_nmf.m:224-231) is formed as W * (H*H'), which
    differs from the reference's expression only where W*H sits at the 1e-9 floor.  Silent rows of V (exact zeros, floored
    to 1e-9 by :169) drive whole rows of W*H to that floor: the result must still meet the solver tolerance."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    F, T, r = 300, 1500, 300
    rs = np.random.default_rng(5)
    V = rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3
    V[:20, :] = 0.0
    V[:, 700:760] = 0.0
    p = dict(cf="ed", sparsity=5, max_iter=8, conv_eps=0, cost_check=1, init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    pl = Plan(gpu_ctx, F, T, r, beta=2.0, max_iter=2, cost_check=True)
    assert "Gram matrix" in pl.describe()
    pl.close()
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
    wr, hr, orf = oracle_nmf(V, p)
    assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH
    assert np.max(np.abs(o["cost"] - orf["cost"]) / np.abs(orf["cost"])) < 1e-5


def test_gram_form_opt_out_keeps_the_floor_of_the_reference(gpu_ctx, monkeypatch):
    """SNMF_GRAM_P=0 forms P = max(W*H, flr) * H' as src/sparse_nmf.m:228-233 writes it.  On data with silent rows AND silent
    frames the rows of W that sit at the floor are where the two forms can differ (by at most flr * sum(h) per entry of P):
    with the opt-out those rows follow the fp64 oracle at the solver tolerance measured on THOSE rows alone; with the Gram
    form the pinned bound is the whole-matrix tolerance (the floor rows carry ~1e-9 of the Frobenius norm)."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    F, T, r = 96, 900, 24
    rs = np.random.default_rng(11)
    V = rs.gamma(0.5, 1.0, (F, 6)) @ rs.gamma(0.3, 1.0, (6, T)) + 1e-3
    V[:10, :] = 0.0
    V[:, 300:360] = 0.0
    p = dict(cf="ed", sparsity=0.5, max_iter=12, conv_eps=0, cost_check=1, init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    wr, hr, orf = oracle_nmf(V, p)
    w_g, h_g, o_g = sparse_nmf(V, p, ctx=gpu_ctx)
    monkeypatch.setenv("SNMF_GRAM_P", "0")
    pl = Plan(gpu_ctx, F, T, r, beta=2.0, max_iter=2, cost_check=True)
    assert "Gram matrix" not in pl.describe()
    pl.close()
    w_f, h_f, o_f = sparse_nmf(V, p, ctx=gpu_ctx)
    monkeypatch.delenv("SNMF_GRAM_P")
    for w, h, o in ((w_g, h_g, o_g), (w_f, h_f, o_f)):
        assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH
        assert np.max(np.abs(o["cost"] - orf["cost"]) / np.abs(orf["cost"])) < 1e-5
    # the floor rows by themselves: the reference's expression tracks the oracle there, row-relative
    assert rel(w_f[:10], wr[:10]) < 1e-3
    # and the Gram form's deviation on them is bounded in ABSOLUTE terms by what the dropped floor can move
    assert np.max(np.abs(w_g[:10] - wr[:10])) < 1e-3 * np.max(np.abs(wr)) + 1e-7


def test_shapes_beyond_the_fused_kernels_take_the_out_of_envelope_path(gpu_ctx):
    """What the fused kernels cannot hold in LDS / registers -- F + r beyond the 16-frame tile images (~2540), W updates
    with r > 1024 under KL or with F = 32n+1 rows -- is no longer refused (src/sparse_nmf.m has no such limit): the plan
    runs the iteration with its intermediates in HBM (csrc/snmf_generic.h; parity in tests/test_gpu_generic.py).  A shape
    inside the envelope keeps the fused kernels."""
    from se_snmf_nat_amd import Plan
    for F, r, kw in ((513, 2100, {}), (512, 1100, {})):
        pl = Plan(gpu_ctx, F, 64, r, beta=1.0, max_iter=2, cost_check=True, **kw)
        assert "out-of-envelope path" in pl.describe()
        pl.close()
    pl = Plan(gpu_ctx, 512, 64, 1100, beta=1.0, max_iter=2, cost_check=True, w_update_ind=np.zeros(1100, bool))  # H-only: fused
    assert "out-of-envelope" not in pl.describe() and "k_hstep" in pl.describe()
    pl.close()


@pytest.mark.parametrize("shape", [(64, 100, 20000, 1.0), (64, 128, 9000, 5.0), (40, 70, 33000, 1.0), (64, 96, 8231, "rvec"), (64, 100, 72000, 5.0),
                                   (48, 65, 300 * 32 + 5, 2.0)], ids=lambda s: "F%d_r%d_T%d" % s[:3])
def test_fused_small_f_iteration_equals_the_two_launches(gpu_ctx, shape, monkeypatch):
    """k_iter_sf (csrc/snmf_smallf.h: the H half-step and the W statistics of a full KL update in one launch, SIMD pairs of an H
    wave and a W wave; run_basis_train.m:90-91 on 64 Mel bands at R = 100) against k_hstep_sf + k_wstats_sf: the same arithmetic
    per tile and the same tiles in the same order on every chunk lane, so with every tile whole (SNMF_HSTEP_SPLIT=0) W and H agree in
    EVERY BIT after several iterations (chunks of 4 R, 4 R + 1 -- the remainder tile through the extra hand-off buffer -- and
    4 R + 2 / 3 tiles), and the objective to the grouping of its fp64 partials.  By default (round 6) a chunk's single remainder tile
    is SHARED by the four pairs -- Lam and Lam' of that tile are then four partial sums added in wave order, another order of
    additions: the default launch agrees with the whole-tile one to the summation-order tolerance on those frames and bit for bit
    on every other frame after ONE iteration (later iterations see the last-bit differences of W everywhere)."""
    from se_snmf_nat_amd import Plan
    F, r, T, sp = shape
    rs = np.random.default_rng(F * 1000 + r)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    sparsity = np.linspace(0.5, 2.0, r) if sp == "rvec" else sp

    def run(fused, iters=4):
        monkeypatch.setenv("SNMF_ITER_SF", "1" if fused else "0")
        pl = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=iters, conv_eps=0.0, cost_check=True, sparsity=sparsity)
        pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
        out = (pl.get_h(np.float32), pl.get_w(), pl.describe(), pl.get_objective())
        pl.close()
        return out

    monkeypatch.delenv("SNMF_HSTEP_SPLIT", raising=False)
    d4, d1 = run(True), run(True, 1)      # the default launch: shared remainder tiles
    monkeypatch.setenv("SNMF_HSTEP_SPLIT", "0")
    a, b, a1 = run(True), run(False), run(True, 1)
    # shared against whole: which frames sit in a shared tile (chunks of 4 R + 1 tiles: the last tile of the chunk)
    n_tiles, nch = (T + 31) // 32, min((T + 31) // 32, 256)
    shared = np.zeros(T, bool)
    if "k_iter_sf" in d4[2]:
        for ch in range(nch):
            tb, te = n_tiles * ch // nch, n_tiles * (ch + 1) // nch
            if (te - tb) % 4 == 1:
                shared[32 * (te - 1):32 * te] = True
    if (T, r) in ((72000, 100), (33000, 70)):
        assert shared.any()  # (these two shapes do have chunks of 4 R + 1 tiles on 256 compute units)
    assert np.array_equal(d1[0][:, ~shared], a1[0][:, ~shared])
    dd = np.abs(d1[0][:, shared] - a1[0][:, shared])
    assert (dd <= 2e-5 * np.abs(a1[0][:, shared]) + 1e-30).all(), dd.max()
    assert np.abs(d4[0] - a[0]).max() <= 1e-4 * np.abs(a[0]).max() and np.abs(d4[1] - a[1]).max() <= 2e-6 * np.abs(a[1]).max()
    for x, y in zip(d4[3][1], a[3][1]):
        assert abs(x - y) <= 1e-6 * abs(y)
    n_cu = int(re.search(r"n_cu=(\d+)", a[2]).group(1)) if "n_cu=" in a[2] else 256
    assert ("k_iter_sf" in a[2]) == ((T + 31) // 32 > n_cu), a[2]
    assert "k_iter_sf" not in b[2]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[3][2] == b[3][2]
    for x, y in zip(a[3][1], b[3][1]):
        assert abs(x - y) <= 1e-9 * abs(y)


@pytest.mark.parametrize("mode", ["full", "semi"])
def test_fused_small_f_iteration_stops_where_the_oracle_stops(gpu_ctx, mode):
    """Early stop (src/sparse_nmf.m:272-284) on the one-launch iteration k_iter_sf: the H step of iteration j + 1 has already run inside
    the launch whose k_wfin fires the stop test of iteration j -- the result must be iterate j's, and the stop index the oracle's."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    F, r, T = 64, 100, 20000
    rs = np.random.default_rng(7)
    V = rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3
    W0, H0 = rs.random((F, r)), rs.random((r, T))
    p = dict(cf="kl", sparsity=1.0, max_iter=120, conv_eps=3e-3, cost_check=1, init_w=W0, init_h=H0)
    kw = {}
    if mode == "semi":
        p["w_update_ind"] = np.arange(r) >= r // 2
        kw["w_update_ind"] = p["w_update_ind"]
    pl = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=120, conv_eps=3e-3, cost_check=True, sparsity=1.0, **kw)
    geo = pl.describe()
    pl.close()
    assert "k_iter_sf" in geo, geo
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
    wr, hr, orf = oracle_nmf(V, p)
    assert 2 < orf["n_iter"] < 120, orf["n_iter"]  # (the case does stop early)
    assert o["n_iter"] == orf["n_iter"]
    assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH
    np.testing.assert_allclose(o["cost"], orf["cost"], rtol=REL_COST)


@pytest.mark.parametrize("F,r,T,mode,spk", [(513, 20, 12000, "full", "scalar"), (513, 30, 21000, "h", "rvec"), (513, 10, 9000, "w", "scalar"),
                                            (257, 32, 26000, "full", "matrix"), (97, 24, 20011, "full", "scalar"), (100, 20, 30000, "semi", "rvec"),
                                            (513, 1, 9000, "full", "scalar"), (385, 17, 8231, "h", "matrix"), (544, 32, 16500, "full", "scalar"),
                                            (129, 5, 40000, "w", "rvec")],
                         ids=lambda v: str(v))
def test_small_rank_family_against_oracle(gpu_ctx, F, r, T, mode, spk):
    """csrc/snmf_smallr.h (round 6): r <= 32 on 3..16 row tiles -- the reference's R = 20 / 10 / 30 settings at F = 513
    (settings/bak_IS16_results/initial_setting_SNMF_Techwin_201603_RT.m:47-48) and r = 32 at F = 257.  k_hstep_sr (a tile per workgroup
    cut by row tiles, the partial numerators meet in LDS one tile late) and k_wstats_sr (a wave owns its rows' statistics for the whole
    chunk) against the fp64 oracle: full, H-only, W-only and partial W updates, every sparsity form (src/sparse_nmf.m:150-155), with and
    without the extra row (F = 32 n + 1), idle waves (3 row tiles), a partial last tile, three to five tiles per workgroup, the maximum
    of sixteen + one row tiles (F = 544 has seventeen -> NOT this family), and the k_wfin row split of such shapes."""
    from se_snmf_nat_amd import Plan, sparse_nmf
    rs = np.random.default_rng(F * 7 + r)
    V = rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3
    W0, H0 = rs.random((F, r)), rs.random((r, T))
    sp = {"scalar": 5.0, "rvec": rs.random(r) * 4, "matrix": rs.random((r, T)) * 3}[spk]
    p = dict(cf="kl", sparsity=sp, max_iter=4, conv_eps=0, cost_check=1, init_w=W0, init_h=H0)
    kw = {}
    if mode == "h":
        p["w_update_ind"] = kw["w_update_ind"] = np.zeros(r, bool)
    elif mode == "w":
        p["h_update_ind"] = kw["h_update_ind"] = np.zeros(r, bool)
    elif mode == "semi":
        p["w_update_ind"] = kw["w_update_ind"] = np.arange(r) >= r // 2
    pl = Plan(gpu_ctx, F, T, r, beta=1.0, max_iter=4, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
    geo = pl.describe()
    pl.close()
    in_family = (F + 31) // 32 <= 16 or F == 513  # 3..16 row tiles (513 = 16 tiles + the extra row)
    assert ("k_hstep_sr" in geo) == (in_family and mode != "w"), geo
    assert ("k_wstats_sr" in geo) == (in_family and mode != "h"), geo
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
    wr, hr, orf = oracle_nmf(V, p)
    assert o["n_iter"] == orf["n_iter"]
    assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH
    np.testing.assert_allclose(o["cost"], orf["cost"], rtol=REL_COST)
    # elementwise too: a slice of a tile finished with the wrong tile's data is a handful of frames, invisible in a norm
    assert np.abs(h - hr).max() <= 2e-4 * np.abs(hr).max()


def test_small_rank_family_stops_where_the_oracle_stops(gpu_ctx):
    """Early stop (src/sparse_nmf.m:272-284) on the small-rank kernels: the device-side flag turns later launches into no-ops, the result
    is the iterate the oracle stops at."""
    from se_snmf_nat_amd import sparse_nmf
    F, r, T = 513, 20, 12000
    rs = np.random.default_rng(11)
    V = rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3
    p = dict(cf="kl", sparsity=5.0, max_iter=60, conv_eps=5e-3, cost_check=1, init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
    wr, hr, orf = oracle_nmf(V, p)  # (stops at iteration 7; its closest approach to the threshold is 5 % away: not a borderline decision)
    assert 2 < orf["n_iter"] < 60 and o["n_iter"] == orf["n_iter"]
    assert rel(w, wr) < REL_WH and rel(h, hr) < REL_WH
    np.testing.assert_allclose(o["cost"], orf["cost"], rtol=REL_COST)
