"""Online separation loop (BASELINE config 3).

CPU tests pin the oracle (oracle/online_oracle.py) against the committed golden run and against
invariants the reference's formulas imply; GPU tests compare the device path (C ABI snmf_online_*,
host mirror se_snmf_nat_amd/online.py) with the oracle.  Tolerances:
  every per-frame decision (iterations of the frame solve, adaptation trigger, sum(r_up), iterations of
  the adaptation solve) must match EXACTLY;
  REL_OUT = 1e-4 Frobenius-relative on the denoised signal before rounding (north_star's fp32 tolerance);
  the int16 stream may differ by at most 1 LSB (a value within fp32 rounding of a .5 boundary).
The loop is a feedback system (activations -> adapted dictionary -> next activations).  Per-atom
activations are the ill-conditioned part of each solve (correlated atoms: the reconstruction B*A is pinned
to ~2e-6 by fp32, single entries of A only to ~1e-5 .. 7e-5, measured), and the adaptation feeds exactly
those back.  With the shipped KL settings the loop stays at ~3e-6 over the whole fixture; the Euclidean
variant (100 un-stopped iterations per frame) drifts to ~7e-4 after 40 frames while every decision still
matches, so that one case carries its own, looser, stated bound.
"""
import os

import numpy as np
import pytest

from oracle.online_oracle import blk_sparse, default_params, frame_stft, ntf_sep_event_rt, synth_ifft_buff

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_OUT = 1e-4


def fixture_inputs(n_hops=None):
    B = np.load(os.path.join(GOLD, "ref_data.npz"))["B"].astype(np.float64)
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"]
    if n_hops is not None:
        s = s[:n_hops * 160]
    rs = np.random.RandomState(1)
    H0 = rs.random_sample(200)
    Ad0 = rs.random_sample((50, 100))
    return s, B[:, :100], B[:, 100:], H0, Ad0


# ---------------------------------------------------------------- CPU: the oracle ------------------
def test_oracle_reproduces_the_golden_run():
    g = np.load(os.path.join(GOLD, "online_is16_124frames.npz"))
    s, Bx, Bd, H0, Ad0 = fixture_inputs()
    o16, of, Bdn, tr = ntf_sep_event_rt(s, Bx, Bd, default_params(), H0, Ad0, return_trace=True)
    assert np.array_equal(o16, g["x_tilde_i16"])
    np.testing.assert_allclose(of, g["x_tilde_f"], rtol=1e-5, atol=1e-3)
    assert np.array_equal([t["n_iter"] for t in tr], g["n_iter"])
    assert np.array_equal([t["adapt_iters"] for t in tr], g["adapt_iters"])
    assert np.array_equal([t["n_up"] for t in tr], g["n_up"])
    np.testing.assert_allclose(Bdn[::4], g["B_DFT_d_sub"], rtol=1e-5, atol=1e-7)


def test_driver_frame_count_delay_and_residue():
    """src/NTF_sep_event_RT.m:67-76,:104: floor(n/hop) full hops + delay+1 zero frames are processed and every
    frame after the first `delay` writes one hop; a trailing partial hop is dropped."""
    p = dict(default_params(), adapt_train_N=0, blk_sparse=0, max_iter=3)
    s, Bx, Bd, H0, Ad0 = fixture_inputs(7)
    s = np.concatenate([s, s[:57]])  # partial hop
    o16, of, _, tr = ntf_sep_event_rt(s, Bx, Bd, p, H0, Ad0, return_trace=True)
    assert len(tr) == 7 + p["delay"] + 1
    assert len(o16) == (len(tr) - p["delay"]) * p["frameshift"]


def test_stft_synthesis_round_trip_is_identity_under_unit_gain():
    """sqrt-Hann analysis * sqrt-Hann synthesis at 75 % overlap with overlapscale = 2*hop/sz sums to one:
    G = 1 must give the input back (away from the edges) -- pins frame_stft + synth_ifft_buff + scale."""
    p = dict(default_params(), DCbin=0, DCbin_back=0, nonzerofloor=0.0)
    rs = np.random.RandomState(3)
    x = rs.randn(160 * 12) * 1000
    sz, hop = p["framelength"], p["frameshift"]
    out = np.zeros(len(x) + sz)
    xp = np.concatenate([np.zeros(sz - hop), x])
    for i in range(len(x) // hop):
        Ym, Yp = frame_stft(xp[i * hop:i * hop + sz], p)
        fr = synth_ifft_buff(Ym, Yp, sz, p["fftlength"], p["win_ISTFT"], 0.0, 0, p["pow"]) * p["overlapscale"]
        out[i * hop:i * hop + sz] += fr
    rec = out[sz - hop:sz - hop + len(x)]
    np.testing.assert_allclose(rec[sz:-sz], x[sz:-sz], rtol=0, atol=1e-6)


def test_blk_sparse_literal_semantics():
    p = default_params()
    rs = np.random.RandomState(0)
    K = 513
    r_blk = rs.rand(K, p["P_len_l"])
    X, D = rs.rand(K) * 1e9, rs.rand(K) * 1e9
    Q, r_out = blk_sparse(X, D, r_blk, 5, p)  # l <= P_len_l: only the initial pattern
    assert np.all(Q[:p["DCbin"]] == 0) and np.all(Q[p["DCbin"]:] == 0.1)
    assert np.array_equal(r_out[:, :-1], r_blk[:, 1:]) and abs(r_out[:, -1].max() - 1.0) < 1e-15
    Q, _ = blk_sparse(X, D, r_blk, 21, p)
    assert np.all(Q[:p["DCbin"]] == 0)
    assert np.all(Q[p["DCbin"]:p["P_len_k"] - 1] == Q[p["P_len_k"] + p["DCbin"] - 1])  # :32
    assert np.all((Q >= 0) & (Q <= 1))
    # a flat block has Hoyer sparsity 0 -> P_val = alpha_p * 0.1
    Qf, _ = blk_sparse(np.ones(K), np.ones(K), np.ones((K, p["P_len_l"])), 21, p)
    assert abs(Qf[200] - p["alpha_p"] * 0.1) < 1e-9


# ---------------------------------------------------------------- GPU: the device path -------------
def _device(s, Bx, Bd, p, H0, Ad0, **kw):
    from se_snmf_nat_amd.online import OnlineSeparator, default_settings
    ps = default_settings()
    ps.update({k: v for k, v in p.items() if k in ps})
    sep = OnlineSeparator(Bx, Bd, ps, H0=H0, Ad_blk0=Ad0, **kw)
    out = sep.process(s, flush=True)
    tr, Bn = sep.trace(), sep.basis()
    sep.close()
    return out, tr, Bn


def _check_trace(tr_dev, n_iter, trig, n_up, adapt_iters):
    assert [t["n_iter"] for t in tr_dev] == list(n_iter)
    assert [t["trig"] for t in tr_dev] == [int(x) for x in trig]
    assert [t["n_up"] for t in tr_dev] == list(n_up)
    assert [t["adapt_iters"] for t in tr_dev] == list(adapt_iters)


@pytest.mark.gpu
def test_device_run_matches_the_golden_run(gpu_ctx):
    g = np.load(os.path.join(GOLD, "online_is16_124frames.npz"))
    s, Bx, Bd, H0, Ad0 = fixture_inputs()
    out, tr, Bn = _device(s, Bx, Bd, default_params(), H0, Ad0, ctx=gpu_ctx)
    _check_trace(tr, g["n_iter"], g["trig"], g["n_up"], g["adapt_iters"])
    ref = g["x_tilde_f"].astype(np.float64)
    assert len(out["x_tilde"]) == len(ref)
    assert np.linalg.norm(out["x_tilde_f"] - ref) / np.linalg.norm(ref) < REL_OUT
    assert np.abs(out["x_tilde"].astype(int) - g["x_tilde_i16"].astype(int)).max() <= 1
    assert np.linalg.norm(Bn[::4] - g["B_DFT_d_sub"]) / np.linalg.norm(g["B_DFT_d_sub"]) < 1e-3
    np.testing.assert_allclose([t["beta"] for t in tr], g["beta"], rtol=1e-3)


@pytest.mark.gpu
def test_adaptation_solve_cooperative_kernel_equals_the_plan_path(gpu_ctx, monkeypatch):
    """The noise-dictionary adaptation (src/bnmf_sep_event_RT_IS16.m:296-336, a W-only sparse_nmf of 513 x 100, rank <= 50)
    runs as ONE cooperative launch (k_wadapt) by default and through the ordinary three-launch plan path with
    SNMF_NO_WADAPT=1 (read when the separator is created): both must take every decision of the golden run -- frame
    iterations, triggers, atoms updated, iterations of each adaptation solve -- and give the same signal and dictionary."""
    g = np.load(os.path.join(GOLD, "online_is16_124frames.npz"))
    s, Bx, Bd, H0, Ad0 = fixture_inputs()
    monkeypatch.delenv("SNMF_NO_WADAPT", raising=False)
    out_c, tr_c, Bn_c = _device(s, Bx, Bd, default_params(), H0, Ad0, ctx=gpu_ctx)
    monkeypatch.setenv("SNMF_NO_WADAPT", "1")
    out_p, tr_p, Bn_p = _device(s, Bx, Bd, default_params(), H0, Ad0, ctx=gpu_ctx)
    for tr in (tr_c, tr_p):
        _check_trace(tr, g["n_iter"], g["trig"], g["n_up"], g["adapt_iters"])
    assert np.linalg.norm(out_c["x_tilde_f"] - out_p["x_tilde_f"]) / np.linalg.norm(out_p["x_tilde_f"]) < REL_OUT
    assert np.linalg.norm(Bn_c - Bn_p) / np.linalg.norm(Bn_p) < 1e-3


VARIANTS = [
    dict(ENHANCE_METHOD="Wiener"),
    dict(blk_sparse=0),
    dict(adapt_train_N=0),
    dict(preemph=0.92, pow=1),
    dict(cf="ed", sparsity=50.0),
    dict(cf="ed", sparsity=50.0, adapt_train_N=0),
    dict(blk_gap=1, P_len_l=4, init_N_len=3),
    dict(conv_eps=0.0, max_iter=12),
    dict(basis_update_N=1, max_iter=30),   # semi-supervised frame solve (:125-127)
    dict(basis_update_E=1, max_iter=30, adapt_train_N=0),
]
REL_OUT_ED_ADAPT = 2e-3  # see the module docstring


@pytest.mark.gpu
@pytest.mark.parametrize("var", VARIANTS, ids=lambda v: "-".join(f"{k}={v[k]}" for k in v))
def test_device_variants_match_the_oracle(gpu_ctx, var):
    p = dict(default_params(), **var)
    s, Bx, Bd, H0, Ad0 = fixture_inputs(36)
    o16, of, Bdn, tr, xh, dh = ntf_sep_event_rt(s, Bx, Bd, p, H0, Ad0, return_trace=True, class_outputs=True)
    out, trd, Bn = _device(s, Bx, Bd, p, H0, Ad0, ctx=gpu_ctx, class_outputs=True)
    _check_trace(trd, [t["n_iter"] for t in tr], [t["trig"] for t in tr], [t["n_up"] for t in tr],
                 [t["adapt_iters"] for t in tr])
    tol = REL_OUT_ED_ADAPT if (var.get("cf") == "ed" and var.get("adapt_train_N", 1)) else REL_OUT
    for dev, ref in ((out["x_tilde_f"], of), (out["x_hat"], xh), (out["d_hat"], dh)):
        assert len(dev) == len(ref)
        ok = np.isfinite(ref)  # a silent tail can be 0/0 in the reference's own formulas (:230): NaN on both sides
        assert np.array_equal(np.isfinite(dev), ok)
        assert np.linalg.norm(dev[ok] - ref[ok]) / max(np.linalg.norm(ref[ok]), 1e-30) < tol
    if tol == REL_OUT:
        assert np.abs(out["x_tilde"].astype(int) - o16.astype(int)).max() <= 1
    assert np.linalg.norm(Bn - Bdn) / np.linalg.norm(Bdn) < 10 * tol


@pytest.mark.gpu
def test_feeding_the_stream_in_chunks_gives_the_same_bits(gpu_ctx):
    from se_snmf_nat_amd.online import default_settings, ntf_sep_event_rt as dev_rt
    s, Bx, Bd, H0, Ad0 = fixture_inputs(40)
    s = np.concatenate([s, s[:33]])  # trailing partial hop
    p = default_settings()
    a16, af, aB = dev_rt(s, Bx, Bd, p, H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx)
    for chunk in (160, 1000, 57):
        b16, bf, bB = dev_rt(s, Bx, Bd, p, H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx, chunk=chunk)
        assert np.array_equal(a16, b16) and np.array_equal(af, bf) and np.array_equal(aB, bB)


@pytest.mark.gpu
def test_unsupported_modes_and_state_errors(gpu_ctx):
    from se_snmf_nat_amd import SnmfError
    from se_snmf_nat_amd.online import OnlineSeparator, default_settings
    s, Bx, Bd, H0, Ad0 = fixture_inputs(4)
    p = default_settings()
    with pytest.raises(ValueError):  # Mel mode needs the Mel dictionaries
        OnlineSeparator(Bx, Bd, dict(p, B_sep_mode="Mel"), ctx=gpu_ctx)
    with pytest.raises(NotImplementedError):
        OnlineSeparator(Bx, Bd, dict(p, Splice=1), ctx=gpu_ctx)
    q = dict(p)
    del q["cost_check"]
    with pytest.raises(KeyError):
        OnlineSeparator(Bx, Bd, q, ctx=gpu_ctx)
    with pytest.raises(SnmfError):
        OnlineSeparator(Bx, Bd, dict(p, blk_gap=2), ctx=gpu_ctx)
    sep = OnlineSeparator(Bx, Bd, p, H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx)
    sep.process(s, flush=True)
    with pytest.raises(SnmfError):
        sep.process(s)
    sep.close()


def _random_setup(seed, fft, sz, hop, R_x, R_d):
    rs = np.random.RandomState(seed)
    F = fft // 2 + 1
    B = rs.gamma(0.6, 1.0, (F, R_x + R_d)) + 1e-3
    B = B / np.sqrt((B ** 2).sum(0)) + 1e-9  # the form run_basis_train.m:113-116 stores
    n = np.arange(sz)
    win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / sz))
    return B[:, :R_x], B[:, R_x:], win


GEOMETRIES = [
    # fft, framelength, hop, R_x, R_d, overrides
    (512, 400, 100, 24, 40, dict(R_a=16, m_a=20, overlap_m_a=0.1, P_len_k=30, P_len_l=6, init_N_len=4, DCbin=3, DCbin_back=3)),
    (512, 512, 128, 32, 32, dict(R_a=32, m_a=40, overlap_m_a=0.05, P_len_k=20, P_len_l=5, init_N_len=2, DCbin=2, DCbin_back=4, blk_gap=5, delay=4)),
    (256, 160, 80, 16, 16, dict(R_a=8, m_a=9, P_len_k=16, P_len_l=3, init_N_len=6, DCbin=1, DCbin_back=1, ENHANCE_METHOD="Wiener",
                                 overlap_m_a=0.25, delay=2)),
    (1024, 640, 160, 50, 70, dict(R_a=50, m_a=30, overlap_m_a=0.05, Ar_up=2.0, beta=2.0, alpha_d=0.8, alpha_eta=0.7, sparsity=1.0)),
    # the reference's exemplar setting R_x = R_d = 500 (settings/bak_IS16_results/initial_setting_Exemplar.m:47-48): r = 1000
    # does not fit the persistent frame-solve kernels; the frame solve runs through the plan loop on 16-frame tiles
    (1024, 640, 160, 500, 500, dict()),
    (1024, 640, 160, 500, 500, dict(adapt_train_N=0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("geo", GEOMETRIES, ids=lambda g: f"fft{g[0]}-sz{g[1]}-hop{g[2]}-r{g[3]}+{g[4]}")
def test_device_other_geometries_match_the_oracle(gpu_ctx, geo):
    """Other frame geometries, dictionary sizes, ring lengths and block-sparsity windows than the shipped ones:
    nothing in the device path may depend on 1024 / 640 / 160 / 100+100 / 50 x 100.  (The ring is kept longer
    than R_a: adapting 32 atoms from a 12-column ring is an under-determined solve whose feedback amplifies
    fp32 rounding to ~3e-5 within 40 frames, enough to move one stop decision -- measured, not a defect.)"""
    fft, sz, hop, R_x, R_d, over = geo
    Bx, Bd, win = _random_setup(fft, fft, sz, hop, R_x, R_d)
    p = dict(default_params(), fftlength=fft, framelength=sz, frameshift=hop, win_STFT=win, win_ISTFT=win.copy(),
             overlapscale=2 * hop / sz)
    p.update(over)
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"][:hop * 44]
    rs = np.random.RandomState(7)
    H0, Ad0 = rs.random_sample(R_x + R_d), rs.random_sample((p["R_a"], p["m_a"]))
    o16, of, Bdn, tr = ntf_sep_event_rt(s, Bx, Bd, p, H0, Ad0, return_trace=True)
    from se_snmf_nat_amd.online import OnlineSeparator, default_settings
    ps = dict(default_settings(), **{k: v for k, v in p.items() if k in default_settings()})
    sep = OnlineSeparator(Bx, Bd, ps, H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx)
    out = sep.process(s, flush=True)
    trd, Bn = sep.trace(), sep.basis()
    sep.close()
    _check_trace(trd, [t["n_iter"] for t in tr], [t["trig"] for t in tr], [t["n_up"] for t in tr], [t["adapt_iters"] for t in tr])
    ok = np.isfinite(of)
    assert np.array_equal(np.isfinite(out["x_tilde_f"]), ok)
    assert np.linalg.norm(out["x_tilde_f"][ok] - of[ok]) / np.linalg.norm(of[ok]) < REL_OUT
    assert np.abs(out["x_tilde"].astype(int) - o16.astype(int)).max() <= 1
    assert np.linalg.norm(Bn - Bdn) / np.linalg.norm(Bdn) < 1e-3
    assert sum(t["solved"] for t in trd) > 0 or not p["adapt_train_N"]


@pytest.mark.gpu
@pytest.mark.parametrize("melconv", [1, 0], ids=["MelConv1", "MelConv0-coupled"])
def test_device_mel_mode_matches_the_oracle(gpu_ctx, melconv):
    """B_sep_mode = 'Mel' (src/bnmf_sep_event_RT_IS16.m:106-120,:165-171,:205-211,:298-318): the frame solve and the
    adaptation run on 64 Mel bands; with MelConv the reconstructions come back through melmat', without it the Mel
    activations drive the DFT bases (coupled dictionaries)."""
    from oracle.frontend_oracle import mel_matrix
    from se_snmf_nat_amd.online import OnlineSeparator, default_settings
    s, Bx, Bd, H0, Ad0 = fixture_inputs(40)
    p = dict(default_params(), B_sep_mode="Mel", MelConv=melconv, F_order=64)
    melmat = mel_matrix(p["fs"], 64, p["fftlength"], 1.0, p["fs"] / 2).T
    BM = melmat @ np.concatenate([Bx, Bd], axis=1)
    BM = BM / np.sqrt((BM ** 2).sum(0)) + 1e-9  # the stored form of run_basis_train.m:115-116
    mel = dict(B_Mel_x=BM[:, :100], B_Mel_d=BM[:, 100:], melmat=melmat)
    o16, of, BMn, tr = ntf_sep_event_rt(s, Bx, Bd, p, H0, Ad0, return_trace=True, mel=mel)
    ps = dict(default_settings(), **{k: v for k, v in p.items() if k in default_settings()})
    sep = OnlineSeparator(Bx, Bd, ps, H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx, B_Mel_x=mel["B_Mel_x"], B_Mel_d=mel["B_Mel_d"])
    out = sep.process(s, flush=True)
    trd, Bm, Bdft = sep.trace(), sep.mel_basis(), sep.basis()
    sep.close()
    _check_trace(trd, [t["n_iter"] for t in tr], [t["trig"] for t in tr], [t["n_up"] for t in tr], [t["adapt_iters"] for t in tr])
    assert sum(t["solved"] for t in trd) > 5
    assert np.linalg.norm(out["x_tilde_f"] - of) / np.linalg.norm(of) < REL_OUT
    assert np.abs(out["x_tilde"].astype(int) - o16.astype(int)).max() <= 1
    assert np.linalg.norm(Bm - BMn) / np.linalg.norm(BMn) < 1e-3
    assert np.array_equal(Bdft.astype(np.float32), Bd.astype(np.float32))  # B_DFT_d is not adapted in Mel mode


@pytest.mark.gpu
def test_long_stream_parity_horizon(gpu_ctx):
    """The adaptive loop amplifies perturbations: on the fixture tiled to 12 s the fp32/fp64 difference grows about
    tenfold per 100 frames (1e-7 at the start, 1e-5 by frame 275, 5e-5 by frame 375) until one stop decision of an
    adaptation solve lands on the other side (frame 413), after which the two trajectories are different, equally
    valid runs (scripts/online_soak.py).  The amplification belongs to the loop: the fp64 oracle run twice with a
    one-ulp change of H0 grows the same way, from nine decades lower (1.7e-16 -> 3e-13 in 500 frames).  What can be
    pinned is the horizon: every decision equal and the signal within 1e-3 over the first 300 frames (3 s), 2.4x
    the length of the committed golden run."""
    from se_snmf_nat_amd.online import OnlineSeparator, default_settings
    s, Bx, Bd, H0, Ad0 = fixture_inputs()
    s = np.tile(s, 3)[:160 * 300]
    o16, of, Bdn, tr = ntf_sep_event_rt(s, Bx, Bd, default_params(), H0, Ad0, return_trace=True)
    sep = OnlineSeparator(Bx, Bd, default_settings(), H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx)
    out = sep.process(s, flush=True)
    trd = sep.trace()
    sep.close()
    _check_trace(trd, [t["n_iter"] for t in tr], [t["trig"] for t in tr], [t["n_up"] for t in tr], [t["adapt_iters"] for t in tr])
    assert np.isfinite(out["x_tilde_f"]).all()
    assert np.linalg.norm(out["x_tilde_f"] - of) / np.linalg.norm(of) < 1e-3
