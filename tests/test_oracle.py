"""CPU tests of the oracle (oracle/sparse_nmf_oracle.py): it is checked against the committed golden
vectors, an independently written loop restatement, scikit-learn's divergence formulas, the
algorithm's invariants and the reference's shipped dictionaries.  W.r.t. MATLAB the parity is a SOFT PIN (the reference
has no tests and cannot run here; its own processed recordings are reproduced in tests/test_refwav.py) -- see
oracle/sparse_nmf_oracle.py."""
import glob
import os

import numpy as np
import pytest

from oracle.sparse_nmf_loops import sparse_nmf_loops
from oracle.sparse_nmf_oracle import OracleError, divergence, run_basis_dnmf_solves, sparse_nmf, synth_problem

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CF = {0.0: "is", 1.0: "kl", 2.0: "ed"}


def load_case(path):
    d = dict(np.load(path))
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    if "V" not in d:
        d["V"] = ref["Y"] if "col" not in d else ref["Y"][:, int(d["col"]):int(d["col"]) + 1]
    if "W0" not in d:
        d["W0"] = ref["B"]
    beta = float(d["beta"])
    p = dict(cf=CF.get(beta, "beta"), beta=beta, sparsity=d["sparsity"] if d["sparsity"].size > 1 else float(d["sparsity"]),
             max_iter=int(d["max_iter"]), conv_eps=float(d["conv_eps"]), init_w=d["W0"].astype(np.float64),
             init_h=d["H0"].astype(np.float64), w_update_ind=d["w_update_ind"], h_update_ind=d["h_update_ind"],
             cost_check=1)
    return d, p


SOLVE_CASES = sorted(f for f in glob.glob(os.path.join(GOLD, "*.npz"))
                     if os.path.basename(f) not in ("ref_data.npz", "dnmf_loop_513x64_r20_20.npz", "frontend_audio.npz",
                                                     "online_is16_124frames.npz", "refwav_pairs.npz",
                                                     "c2_full_257x100000_r256.npz", "basis_v73_small_expected.npz"))


@pytest.mark.parametrize("path", SOLVE_CASES, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_golden(path):
    d, p = load_case(path)
    w, h, o = sparse_nmf(d["V"].astype(np.float64), p)
    assert o["n_iter"] == int(d["n_iter"])
    if "W" in d:
        np.testing.assert_allclose(w, d["W"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(h, d["H"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(o["cost"], d["cost"], rtol=1e-12)
    np.testing.assert_allclose(o["div"], d["div"], rtol=1e-12)


def test_oracle_reproduces_golden_dnmf_loop():
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    d = dict(np.load(os.path.join(GOLD, "dnmf_loop_513x64_r20_20.npz")))
    Y = ref["Y"]
    X = (Y * d["mask"] + 1e-9).astype(np.float32)
    D = (Y - X + 2e-9).astype(np.float32)
    Bs = np.concatenate([ref["B"][:, :20], ref["B"][:, 100:120]], axis=1)
    p = dict(cf="kl", sparsity=5, max_iter=30, conv_eps=1e-3, cost_check=1)
    B_hat, A_hat = run_basis_dnmf_solves(*(a.astype(np.float64) for a in (Y, X, D, Bs)), 20, 20, p)
    np.testing.assert_allclose(B_hat, d["B_hat"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(A_hat, d["A_hat"], rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("beta", [0.0, 0.5, 1.0, 1.5, 2.0])
@pytest.mark.parametrize("masks", ["full", "semi", "wonly", "honly"])
def test_vectorised_oracle_matches_loop_restatement(beta, masks):
    V, W0, H0 = synth_problem(8, 6, 3, seed_data=5, seed_init=6)
    wi = {"full": [1, 1, 1], "semi": [0, 1, 1], "wonly": [1, 1, 1], "honly": [0, 0, 0]}[masks]
    hi = [0, 0, 0] if masks == "wonly" else [1, 1, 1]
    p = dict(cf=CF.get(beta, "beta"), beta=beta, sparsity=0.3, max_iter=5, conv_eps=0, init_w=W0, init_h=H0,
             w_update_ind=np.array(wi, bool), h_update_ind=np.array(hi, bool), cost_check=1)
    w, h, o = sparse_nmf(V, p)
    w2, h2, d2, c2 = sparse_nmf_loops(V.tolist(), W0.tolist(), H0.tolist(), beta, 0.3, 5, 0, wi, hi)
    np.testing.assert_allclose(w, np.array(w2), rtol=1e-12)
    np.testing.assert_allclose(h, np.array(h2), rtol=1e-12)
    np.testing.assert_allclose(o["cost"], c2, rtol=1e-12)
    np.testing.assert_allclose(o["div"], d2, rtol=1e-12)


@pytest.mark.parametrize("beta", [0.0, 0.5, 1.0, 1.5, 2.0, 3.0])
def test_divergence_matches_sklearn(beta):
    from sklearn.decomposition._nmf import _beta_divergence
    rs = np.random.RandomState(0)
    V = rs.gamma(1.0, 1.0, (13, 11)) + 1e-3
    W = rs.random_sample((13, 4)) + 0.1
    H = rs.random_sample((4, 11)) + 0.1
    ours = divergence(V, W @ H, beta)
    sk = _beta_divergence(V, W, H, beta, square_root=False)
    if beta == 2.0:
        sk *= 2.0  # sklearn uses 1/2 ||.||^2, the reference sum((v-lam).^2) (src/sparse_nmf.m:252)
    np.testing.assert_allclose(ours, sk, rtol=1e-9)


@pytest.mark.parametrize("beta,cf", [(0.0, "is"), (0.5, "x"), (1.0, "kl"), (2.0, "ed")])
def test_cost_is_monotone_and_w_unit_norm(beta, cf):
    V, W0, H0 = synth_problem(65, 200, 12)
    w, h, o = sparse_nmf(V, dict(cf=cf, beta=beta, sparsity=1.0, max_iter=40, init_w=W0, init_h=H0, cost_check=1))
    assert np.all(np.diff(o["cost"]) <= 1e-9 * np.abs(o["cost"][:-1]))
    np.testing.assert_allclose(np.sqrt((w ** 2).sum(0)), 1.0, rtol=1e-12)
    assert (w >= 0).all() and (h >= 0).all()


def test_exact_factorisation_is_a_fixed_point():
    rs = np.random.RandomState(2)
    W = rs.random_sample((20, 4)) + 0.1
    W /= np.sqrt((W ** 2).sum(0))
    H = rs.random_sample((4, 30)) + 0.1
    for cf in ("kl", "ed", "is"):
        w, h, o = sparse_nmf(W @ H, dict(cf=cf, sparsity=0, max_iter=3, init_w=W, init_h=H, cost_check=1))
        np.testing.assert_allclose(w, W, rtol=1e-9)
        np.testing.assert_allclose(h, H, rtol=1e-9)
        assert abs(o["div"][-1]) < 1e-9


def test_initial_normalisation_makes_column_scale_irrelevant():
    V, W0, H0 = synth_problem(30, 40, 5)
    s = np.array([1.0, 10.0, 0.1, 3.0, 7.0])
    a = sparse_nmf(V, dict(sparsity=1, max_iter=5, init_w=W0, init_h=H0, cost_check=1))
    b = sparse_nmf(V, dict(sparsity=1, max_iter=5, init_w=W0 * s, init_h=H0 / s[:, None], cost_check=1))
    np.testing.assert_allclose(a[0], b[0], rtol=1e-10)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-10)


def test_masks_and_early_stop_semantics():
    V, W0, H0 = synth_problem(30, 40, 6)
    base = dict(sparsity=1, max_iter=8, init_w=W0, init_h=H0, cost_check=1)
    wn = W0 / np.sqrt((W0 ** 2).sum(0))
    w, h, _ = sparse_nmf(V, dict(base, w_update_ind=np.zeros(6, bool)))
    np.testing.assert_allclose(w, wn, rtol=1e-12)  # H-only keeps the normalised dictionary
    w, h, _ = sparse_nmf(V, dict(base, h_update_ind=np.zeros(6, bool)))
    np.testing.assert_allclose(h, H0 * np.sqrt((W0 ** 2).sum(0))[:, None], rtol=1e-12)  # W-only: rescaled init_h
    w, h, _ = sparse_nmf(V, dict(base, w_update_ind=np.arange(6) >= 3))
    np.testing.assert_allclose(w[:, :3], wn[:, :3], rtol=1e-12)
    # early stop: vectors truncated to the stop iteration, which is > 1
    w, h, o = sparse_nmf(V, dict(base, max_iter=500, conv_eps=1e-2))
    assert 1 < o["n_iter"] < 500 and len(o["cost"]) == o["n_iter"] == len(o["div"])
    e = abs(o["cost"][-1] - o["cost"][-2]) / o["cost"][-2]
    assert e < 1e-2
    # without cost_check: zeros(1, max_iter), never stops
    w, h, o = sparse_nmf(V, dict(base, conv_eps=1e-2, cost_check=0))
    assert o["n_iter"] == 8 and not o["cost"].any() and len(o["cost"]) == 8


def test_reference_error_behaviour():
    V, W0, H0 = synth_problem(10, 12, 3)
    with pytest.raises(OracleError, match="Number of components or initialization must be given"):
        sparse_nmf(V, dict(cost_check=1))
    with pytest.raises(OracleError, match="cost_check"):
        sparse_nmf(V, dict(r=3))  # src/sparse_nmf.m:260 has no default
    with pytest.raises(OracleError):
        sparse_nmf(V, dict(init_w=W0, init_h=H0, h_update_ind=np.array([1, 0, 1], bool), cost_check=1))
    # r larger than init_w: random columns appended (:125-127)
    w, h, _ = sparse_nmf(V, dict(init_w=W0, r=5, max_iter=2, cost_check=1))
    assert w.shape == (10, 5) and h.shape == (5, 12)
    w, h, _ = sparse_nmf(V, dict(init_w=W0, init_h="ones", max_iter=2, cost_check=1))
    assert h.shape == (3, 12)


def test_display_lines_have_the_reference_format(capsys):
    """p.display ~= 0 (src/sparse_nmf.m:162-164 default 0): header :181-183, one erased-and-rewritten line per iteration
    inside `if p.cost_check` :266-270, the convergence line :276-278, and the closing disp of a SINGLE-quoted string
    :288-290 whose backslash-n stay literal and which is printed after a convergence stop as well."""
    import re
    V, W0, H0 = synth_problem(33, 80, 5)
    p = dict(cf="kl", sparsity=1, max_iter=300, conv_eps=1e-2, init_w=W0, init_h=H0, cost_check=1, display=1)
    _, _, o = sparse_nmf(V, p)
    out = capsys.readouterr().out
    assert 2 < o["n_iter"] < 300
    assert out.startswith("Performing sparse NMF with beta-divergence, beta=1.0\n")
    lines = re.findall(r"iteration (\d+) div = (\d\.\d{3}e[+-]\d{2}) cost = (\d\.\d{3}e[+-]\d{2})", out)
    assert [int(x[0]) for x in lines] == list(range(1, o["n_iter"] + 1))
    assert float(lines[-1][2]) == pytest.approx(o["cost"][-1], rel=1e-3)
    first = "iteration 1 div = %.3e cost = %.3e" % (o["div"][0], o["cost"][0])
    assert first + "\b" * len(first) + "iteration 2" in out  # the previous line is erased with backspaces
    assert out.endswith("Convergence reached, aborting iteration\n\\nMax Iteration reached, aborting iteration\\n\n")
    # without cost_check: header and closing line only; display = 0 (the default): nothing
    sparse_nmf(V, dict(p, cost_check=0, max_iter=3))
    assert capsys.readouterr().out == "Performing sparse NMF with beta-divergence, beta=1.0\n\\nMax Iteration reached, aborting iteration\\n\n"
    sparse_nmf(V, dict(p, display=0, max_iter=3))
    assert capsys.readouterr().out == ""
    # src/sparse_nmf_GPU.m:266-268: one plain line per iteration
    sparse_nmf(V, dict(p, max_iter=2, conv_eps=0), gpu_variant=True)
    g = capsys.readouterr().out.splitlines()
    assert len(g) == 2 and g[0].startswith("iteration 1 div = ") and g[1].startswith("iteration 2 div = ")


def test_gpu_variant_deltas():
    V, W0, H0 = synth_problem(20, 30, 4)
    p = dict(sparsity=1, max_iter=6, init_w=W0, init_h=H0)
    w, h, o = sparse_nmf(V, p, gpu_variant=True)  # no cost_check needed, objective left zero
    assert not o["div"].any() and len(o["div"]) == 6 and not o["cost"].any()
    w2, h2, _ = sparse_nmf(V, dict(p, cost_check=1))  # V >= 1e-9 everywhere: the floor is the only other delta
    np.testing.assert_array_equal(w, w2)
    np.testing.assert_array_equal(h, h2)
    # the convergence test still runs in the GPU file (sparse_nmf_GPU.m:270-277)
    _, _, o3 = sparse_nmf(V, dict(p, max_iter=500, conv_eps=1e-2), gpu_variant=True)
    _, _, o4 = sparse_nmf(V, dict(p, max_iter=500, conv_eps=1e-2, cost_check=1))
    assert o3["n_iter"] == o4["n_iter"] < 500 and len(o3["div"]) == 500


def test_sparsity_forms_agree():
    V, W0, H0 = synth_problem(20, 30, 4)
    base = dict(max_iter=5, init_w=W0, init_h=H0, cost_check=1)
    a = sparse_nmf(V, dict(base, sparsity=2.0))
    b = sparse_nmf(V, dict(base, sparsity=np.full((4, 1), 2.0)))
    c = sparse_nmf(V, dict(base, sparsity=np.full((4, 30), 2.0)))
    for x in (b, c):
        np.testing.assert_allclose(a[0], x[0], rtol=1e-13)
        np.testing.assert_allclose(a[1], x[1], rtol=1e-13)
        np.testing.assert_allclose(a[2]["cost"], x[2]["cost"], rtol=1e-13)


def test_shipped_dictionaries_have_the_training_output_format():
    """run_basis_train.m:113-116: columns normalised to unit L2 norm, then + 1e-9."""
    ref = np.load(os.path.join(GOLD, "ref_data.npz"))
    for key in ("B", "Bu"):
        B = ref[key].astype(np.float64)
        assert B.min() >= 0.99e-9
        nrm = np.sqrt(((B - 1e-9) ** 2).sum(0))
        np.testing.assert_allclose(nrm, 1.0, atol=2e-6)


def test_full_size_c2_golden_is_consistent_with_the_oracle_on_a_frame_block():
    """The full-size golden (tests/golden/make_golden_c2.py: 12 minutes of oracle time) cannot be regenerated in a
    CPU test, but it can be checked for consistency in seconds: H-only updates are frame-local given W, so running
    the oracle's H step on the golden's W12 over the first 64 frames of the SAME inputs must leave the stored
    H12_head at a point where one more H-only iteration moves it exactly as the full solve would -- checked through
    the cheap invariants instead: costs strictly decreasing, 12-iteration vectors equal to the head of the long run,
    W12 unit-norm and non-negative, H slices non-negative and finite."""
    g = np.load(os.path.join(GOLD, "c2_full_257x100000_r256.npz"))
    assert g["W12"].shape == (257, 256) and g["H12_head"].shape == (256, 64) and g["H12_tail"].shape == (256, 64)
    np.testing.assert_array_equal(g["cost"][:12], g["cost12"])
    np.testing.assert_array_equal(g["div"][:12], g["div12"])
    assert np.all(np.diff(g["cost"]) < 0) and np.all(g["div"] < g["cost"])
    np.testing.assert_allclose(np.sqrt((g["W12"] ** 2).sum(0)), 1.0, rtol=1e-12)
    assert (g["W12"] >= 0).all() and (g["H12_head"] >= 0).all() and np.isfinite(g["H12_tail"]).all()
    # the first 64 frames after ONE oracle iteration from the same start equal a 64-frame H-only solve with W fixed at
    # its normalised start (frame locality of src/sparse_nmf.m:189-195), which pins make_problem's block generator
    from bench import F_, R_, SPARSITY, make_problem
    V, W0, H0 = make_problem(F_, 2000, R_)  # the first two 1000-frame blocks: identical to the head of the full problem
    V = V.astype(np.float32).astype(np.float64)
    H0 = H0.astype(np.float32).astype(np.float64)
    _, h1, o1 = sparse_nmf(V[:, :64], dict(cf="kl", sparsity=SPARSITY, max_iter=1, conv_eps=0, init_w=W0, init_h=H0[:, :64],
                                          cost_check=1, w_update_ind=np.zeros(R_, bool)))
    _, h2, _ = sparse_nmf(V, dict(cf="kl", sparsity=SPARSITY, max_iter=1, conv_eps=0, init_w=W0, init_h=H0, cost_check=1,
                                  w_update_ind=np.zeros(R_, bool)))
    np.testing.assert_allclose(h1, h2[:, :64], rtol=1e-12)


REF_ROOT = os.environ.get("SNMF_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_ROOT, "basis")), reason="reference tree not present (GPU box)")
def test_shipped_basis_files_load_through_the_product_reader_and_round_trip(tmp_path):
    """The dictionaries the reference ships (basis/*/R_100.mat, B_D_u.mat: MAT-5 files written by run_basis_train.m:136
    and src/NTF_sep_event_RT.m:138-140) load through se_snmf_nat_amd.train.load_basis_mat with the reference's variable
    names; they have the form run_basis_train.m:113-116 stores (unit-L2 columns + 1e-9); the committed fp32 fixture
    tests/golden/ref_data.npz is their rounding; and save_basis_mat writes a file the same reader (and MATLAB's load)
    reads back bit for bit."""
    from se_snmf_nat_amd.train import load_basis_mat, save_basis_mat
    ref = np.load(os.path.join(GOLD, "ref_data.npz"))
    sub = "TASLP_Splice0-SNMF_p2_DD0"
    for k, (cls, cols) in enumerate((("Clean_train_TIMIT_test", slice(0, 100)), ("CHiME3_bgn_ch6", slice(100, 200)))):
        m = load_basis_mat(os.path.join(REF_ROOT, "basis", cls, sub, "R_100.mat"))
        assert {"B_DFT_sub", "B_Mel_sub"} <= set(m)  # the shipped files carry the two dictionaries (the A_* of :136 were not shipped)
        B = np.asarray(m["B_DFT_sub"], dtype=np.float64)
        assert B.shape == (513, 100) and m["B_Mel_sub"].shape[1] == 100
        np.testing.assert_allclose(np.sqrt(((B - 1e-9) ** 2).sum(0)), 1.0, rtol=1e-6)  # :113-114
        np.testing.assert_array_equal(B.astype(np.float32), ref["B"][:, cols])
    u = load_basis_mat(os.path.join(REF_ROOT, "B_D_u.mat"))
    assert {"B_DFT_d", "B_Mel_d"} <= set(u)
    np.testing.assert_array_equal(np.asarray(u["B_DFT_d"])[:, :50].astype(np.float32), ref["Bu"])
    np.testing.assert_array_equal(u["B_DFT_d"], u["B_Mel_d"])  # DFT mode: the Mel slots hold the DFT bases
    out = {"B_DFT_sub": B, "B_Mel_sub": np.asarray(m["B_Mel_sub"], float), "A_DFT_sub": 0, "A_Mel_sub": 0}
    fn = str(tmp_path / "R_100.mat")
    save_basis_mat(fn, out)
    back = load_basis_mat(fn)
    np.testing.assert_array_equal(back["B_DFT_sub"], B)
    np.testing.assert_array_equal(back["B_Mel_sub"], out["B_Mel_sub"])
