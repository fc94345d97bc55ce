"""Spectrogram front-end: oracle checks on CPU, HIP parity on the GPU."""
import os

import numpy as np
import pytest

from oracle import frontend_oracle as fo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_stft_oracle_matches_a_direct_dft_and_the_frame_count_rule():
    rs = np.random.RandomState(0)
    p = fo.default_params()
    s = rs.randn(3000) * 1000
    S = fo.stft_fft(s, p["framelength"], p["frameshift"], p["fftlength"], p["DCbin"], p["win_STFT"], 0.5)
    assert S.shape == (513, 3000 // 160)
    nproc = int(np.any(S != 0, axis=0).sum())
    assert nproc == int(np.ceil((3000 - 1024 - 1) / 160))  # while size_crnt < length(s) - fftlen
    n = np.arange(1024)
    for i in (0, 3, nproc - 1):
        x = s[i * 160:i * 160 + 640].copy()
        y = x.copy()
        y[1:] -= 0.5 * x[:-1]
        y *= p["win_STFT"]
        for f in (7, 100, 512):
            X = np.sum(y * np.exp(-2j * np.pi * f * n[:640] / 1024))
            np.testing.assert_allclose(S[f, i], abs(X), rtol=1e-10)
        assert np.all(S[:5, i] == 1e-6)
    assert not S[:, nproc:].any()
    assert fo.stft_fft(rs.randn(1000), 640, 160, 1024, 5, p["win_STFT"], 0.0).any() == False  # too short


def test_frame_splice_and_mel_matrix_oracles():
    F = np.arange(12.0).reshape(3, 4) + 1
    S = fo.frame_splice(F, 1)
    assert S.shape == (9, 4)
    np.testing.assert_array_equal(S[3:6], F)                       # centre block = the frame itself
    np.testing.assert_array_equal(S[6:9, :3], F[:, 1:])            # upper block = next frame
    np.testing.assert_array_equal(S[0:3, 1:], F[:, :3])            # lower block = previous frame
    assert not S[0:3, 0].any() and not S[6:9, 3].any()             # zero outside the signal
    np.testing.assert_array_equal(fo.frame_splice(F, 0), F)
    M = fo.mel_matrix(16000, 64, 1024)
    assert M.shape == (513, 64) and M.min() >= 0 and np.isclose(M.max(), 1.0)
    peaks = M.argmax(0)
    assert np.all(np.diff(peaks) > 0)                              # centres increase with the channel
    # Entries worked out BY HAND from src/mel_matrix.m:16-38 for (fs, NbCh, Nfft) = (16000, 64, 1024) -- the product-side
    # table and the oracle's are near-identical restatements, so comparing them with each other would pin nothing:
    #   LowMel = 2595*log10(1+64/700) = 98.60, NyqMel = 2595*log10(1+8000/700) = 2840.02
    #   channel 1 : fCen = 64 Hz     -> StartBin = round(4.096)+1 = 5;  StartBin(2) = round(5.960)+1 = 7 -> LowLen = 3;
    #               EndMel -> 123.39 Hz -> EndBin = round(7.897)+1 = 9 -> TotLen = 5, HiLen = 3
    #               rows 5..7 = (1:3)/3, rows 7..9 = (3:-1:1)/3
    #   channel 32: StartBin = 112 (1737.40 Hz), StartBin(33) = 118 -> LowLen = 7; EndBin = 124 -> HiLen = 7
    #   channel 64: StartBin = 473 (7372.61 Hz), EndBin(63) = 493 -> LowLen = 21; EndBin = 513 -> TotLen = 41, HiLen = 21
    from se_snmf_nat_amd.frontend import mel_matrix
    for name, tab in (("oracle", M), ("product", mel_matrix(16000, 64, 1024))):
        def col(k):  # 1-based channel -> (first bin, last bin, weights), 1-based bins like the reference
            nz = np.nonzero(tab[:, k - 1])[0]
            return nz[0] + 1, nz[-1] + 1, tab[nz[0]:nz[-1] + 1, k - 1]
        b0, b1, wts = col(1)
        assert (b0, b1) == (5, 9), name
        np.testing.assert_allclose(wts, [1 / 3, 2 / 3, 1, 2 / 3, 1 / 3], rtol=1e-15, err_msg=name)
        b0, b1, wts = col(32)
        assert (b0, b1) == (112, 124), name
        np.testing.assert_allclose(wts, np.r_[np.arange(1, 8), np.arange(6, 0, -1)] / 7, rtol=1e-15, err_msg=name)
        b0, b1, wts = col(64)
        assert (b0, b1) == (473, 513), name
        np.testing.assert_allclose(wts, np.r_[np.arange(1, 22), np.arange(20, 0, -1)] / 21, rtol=1e-15, err_msg=name)


def test_oracle_reproduces_frontend_golden():
    g = dict(np.load(os.path.join(GOLD, "frontend_audio.npz")))
    p = fo.default_params()
    V = fo.dft_features(g["samples"].astype(np.float64), p)
    assert V.shape[1] == int(g["n_frames"]) and V.min() >= 1e-9
    np.testing.assert_allclose(V[::8, ::4], g["V_sub"], rtol=1e-12)
    np.testing.assert_allclose(fo.mel_features(V, p)[::4, ::4], g["mel_sub"], rtol=1e-12)
    V1 = fo.dft_features(g["samples"].astype(np.float64), dict(p, Splice=1, preemph=0.92, pow=1))
    np.testing.assert_allclose(V1[::16, ::4], g["V1_sub"], rtol=1e-12)


# ---------------------------------------------------------------------------------- GPU parity
# Tolerance: an fp32 radix-2 FFT of 1024 points carries an absolute error of ~eps_f32*log2(N)*||x||
# per bin, i.e. relative to the LARGEST bin of the frame; powers square it.
def _close(gpu, ref, colmax):
    err = np.abs(gpu - ref)
    assert np.all(err <= 1e-4 * np.abs(ref) + 2e-6 * colmax[None, :]), float((err / (np.abs(ref) + 1e-30)).max())


@pytest.mark.gpu
def test_stft_features_on_device_match_the_oracle(gpu_ctx):
    from se_snmf_nat_amd import frontend as fe
    g = dict(np.load(os.path.join(GOLD, "frontend_audio.npz")))
    s = g["samples"].astype(np.float64)
    for over in (dict(), dict(Splice=1, preemph=0.92, pow=1), dict(Splice=2, pow=0.7, DCbin=1),
                 dict(framelength=400, frameshift=100, fftlength=512,
                      win_STFT=np.hanning(400), DCbin=3)):
        p = dict(fo.default_params(), **over)
        ref = fo.dft_features(s, p)
        out = fe.stft_features(s, p, ctx=gpu_ctx)
        assert out.shape == ref.shape and fe.num_frames(len(s), p) == ref.shape[1]
        K = p["fftlength"] // 2 + 1
        colmax = ref.reshape(-1, K, ref.shape[1]).max(axis=(0, 1)) if p["Splice"] else ref.max(0)
        colmax = np.maximum(colmax, ref.max() * 1e-3)  # spliced neighbours may dominate a column
        _close(out.astype(np.float64), ref, colmax)
    # Mel projection
    p = fo.default_params()
    V = fo.dft_features(s, p)
    mel = fe.mel_features(V, p, ctx=gpu_ctx)
    np.testing.assert_allclose(mel, fo.mel_features(V.astype(np.float32).astype(np.float64), p), rtol=2e-5)
    assert fe.num_frames(1000, p) == 0 and fe.stft_features(np.zeros(1000), p, ctx=gpu_ctx).shape == (513, 0)


@pytest.mark.gpu
def test_audio_to_activations_without_v_crossing_pcie(gpu_ctx):
    """audio -> features in HBM -> H-only solve with the shipped dictionaries, vs the oracle chain
    (src/bnmf_sep_event_RT_IS16.m:67-78,138-154 shape, batched over a file)."""
    from oracle.sparse_nmf_oracle import sparse_nmf as oracle_nmf
    from se_snmf_nat_amd import Plan, frontend as fe
    g = dict(np.load(os.path.join(GOLD, "frontend_audio.npz")))
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    s = g["samples"].astype(np.float64)
    p = fo.default_params()
    B = ref["B"].astype(np.float64)
    T = fe.num_frames(len(s), p)
    H0 = np.random.RandomState(3).random_sample((200, T))
    plan = Plan(gpu_ctx, 513, T, 200, beta=1.0, max_iter=25, conv_eps=0.0, cost_check=True, sparsity=5.0,
                w_update_ind=np.zeros(200, bool))
    fe.set_plan_v_from_audio(plan, s, p)
    plan.set_w(B); plan.set_h(H0); plan.init(); plan.run()
    h = plan.get_h()
    div, cost, n = plan.get_objective()
    V = fo.dft_features(s, p)
    wr, hr, orf = oracle_nmf(V, dict(cf="kl", sparsity=5, max_iter=25, init_w=B, init_h=H0, cost_check=1,
                                     w_update_ind=np.zeros(200, bool)))
    assert np.linalg.norm(h - hr) / np.linalg.norm(hr) < 1e-4
    np.testing.assert_allclose(cost[:n], orf["cost"], rtol=2e-5)


@pytest.mark.gpu
def test_basis_training_caller_end_to_end(gpu_ctx, tmp_path):
    """run_basis_train.m:58-136 on the GPU (features, Mel, two full-update solves, renormalise, save)
    against the oracle chain; the saved file has the format of the reference's shipped R_100.mat."""
    from se_snmf_nat_amd import train
    g = dict(np.load(os.path.join(GOLD, "frontend_audio.npz")))
    s = g["samples"].astype(np.float64)
    p = dict(fo.default_params(), cf="kl", sparsity=5, max_iter=20, conv_eps=0, cost_check=1, cluster_buff=1,
             train_Exemplar=0)
    idx = np.random.RandomState(5).choice(114, size=16, replace=False) + 1
    out = train.run_basis_train_signal(s, 16, p, sample_idx=idx, ctx=gpu_ctx)
    ref = fo.run_basis_train_signal(s, 16, p, idx)
    for k in ("B_DFT_sub", "B_Mel_sub", "A_DFT_sub", "A_Mel_sub"):
        assert out[k].shape == ref[k].shape
        assert np.linalg.norm(out[k] - ref[k]) / np.linalg.norm(ref[k]) < 2e-4, k
    assert out["B_DFT_sub"].min() >= 1e-9 and out["B_DFT_sub"].shape == (513, 16) and out["B_Mel_sub"].shape == (64, 16)
    np.testing.assert_allclose(np.sqrt(((out["B_DFT_sub"] - 1e-9) ** 2).sum(0)), 1.0, atol=1e-6)
    f = str(tmp_path / "R_16.mat")
    train.save_basis_mat(f, out)
    back = train.load_basis_mat(f)
    assert set(back) == {"B_DFT_sub", "B_Mel_sub", "A_DFT_sub", "A_Mel_sub"}
    np.testing.assert_array_equal(back["B_DFT_sub"], out["B_DFT_sub"])


def test_tf_dd_oracle_is_the_recursive_average():
    """oracle/frontend_oracle.py::tf_dd against two independent statements of src/TF_DD.m:1-9: scipy's IIR filter run
    with the first column as its initial state, and the closed form X_DD(:,l) = a^(l-1) X(:,1) + (1-a) sum_j a^(l-j) X(:,j)."""
    from scipy.signal import lfilter, lfiltic
    rs = np.random.RandomState(2)
    X = rs.gamma(0.5, 1.0, (7, 300))
    for a in (0.4, 0.95, 0.0):  # 0.4: settings/initial_setting_SNMF_NAT.m:119
        ref = np.empty_like(X)
        for f in range(X.shape[0]):
            zi = lfiltic([1 - a], [1, -a], y=[X[f, 0]])
            ref[f, 0] = X[f, 0]
            ref[f, 1:] = lfilter([1 - a], [1, -a], X[f, 1:], zi=zi)[0]
        got = fo.tf_dd(X, {"alpha_eta": a})
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=0)
        l = 37
        closed = a ** l * X[:, 0] + (1 - a) * sum(a ** (l - j) * X[:, j] for j in range(1, l + 1))
        np.testing.assert_allclose(got[:, l], closed, rtol=1e-10)
    np.testing.assert_array_equal(fo.tf_dd(X[:, :1], {"alpha_eta": 0.4}), X[:, :1])  # one frame: X_DD = X


@pytest.mark.gpu
@pytest.mark.parametrize("shape,alpha", [((513, 3000), 0.4), ((64, 257), 0.95), ((513, 256), 0.4), ((1, 1000), 0.7), ((700, 1), 0.4),
                                         ((257, 100000), 0.98)])
def test_tf_dd_on_device_against_oracle(gpu_ctx, shape, alpha):
    """src/TF_DD.m on the GPU (chunked scan, csrc/snmf_frontend.h) against the fp64 oracle: fp32 in and out, fp64 state --
    within fp32 rounding of the result (tolerance 2e-7 relative + a hair of the row scale), including chunk boundaries
    (multiples of 256 frames), one row, one frame and a slowly forgetting average over 100 000 frames."""
    from se_snmf_nat_amd import frontend as fe
    rs = np.random.RandomState(shape[0] + shape[1])
    X = (rs.gamma(0.5, 1.0, shape) * 10.0 ** rs.uniform(-3, 3, (shape[0], 1))).astype(np.float32)
    got = fe.tf_dd(X, {"alpha_eta": alpha}, ctx=gpu_ctx)
    ref = fo.tf_dd(X.astype(np.float64), {"alpha_eta": alpha})
    assert got.dtype == np.float32 and got.shape == X.shape
    assert (np.abs(got - ref) <= 2e-7 * np.abs(ref) + 1e-7 * np.abs(ref).max(axis=1, keepdims=True)).all()
    np.testing.assert_array_equal(got[:, 0], X[:, 0])


@pytest.mark.gpu
def test_basis_training_with_domain_dd(gpu_ctx):
    """run_basis_train.m:64-67: with p.domain_DD the features are replaced by TF_DD(TF_mag, p) BEFORE the Mel projection
    and both solves; against the oracle chain with the same flag, and different from the run without it."""
    from se_snmf_nat_amd import train
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"].astype(np.float64)
    p = dict(fo.default_params(), cf="kl", sparsity=5, max_iter=20, conv_eps=0, cost_check=1, cluster_buff=1,
             train_Exemplar=0, domain_DD=1, alpha_eta=0.4)
    idx = np.random.RandomState(5).choice(114, size=16, replace=False) + 1
    out = train.run_basis_train_signal(s, 16, p, sample_idx=idx, ctx=gpu_ctx)
    ref = fo.run_basis_train_signal(s, 16, p, idx)
    plain = fo.run_basis_train_signal(s, 16, dict(p, domain_DD=0), idx)
    for k in ("B_DFT_sub", "B_Mel_sub", "A_DFT_sub", "A_Mel_sub"):
        assert np.linalg.norm(out[k] - ref[k]) / np.linalg.norm(ref[k]) < 2e-4, k
    assert np.linalg.norm(ref["B_DFT_sub"] - plain["B_DFT_sub"]) / np.linalg.norm(plain["B_DFT_sub"]) > 1e-2


@pytest.mark.gpu
def test_basis_training_with_kmeans_rank_reduction(gpu_ctx):
    """run_basis_train.m:118-129 (cluster_buff > 1): the dictionary is trained with cluster_buff*R atoms and reduced to
    R by clustering the Mel atoms; the SAME atoms are kept in both dictionaries and both activation matrices.  Checked
    against the un-reduced run (cluster_buff = 1, R' = cluster_buff*R, same exemplar columns) + the loop restatement of
    the clustering on that run's Mel dictionary."""
    from oracle import kmeans_oracle
    from se_snmf_nat_amd import train
    from se_snmf_nat_amd.api import SnmfError
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"].astype(np.float64)
    base = dict(fo.default_params(), cf="kl", sparsity=5, max_iter=15, conv_eps=0, cost_check=1, train_Exemplar=0)
    idx = np.random.RandomState(5).choice(114, size=16, replace=False) + 1
    full = train.run_basis_train_signal(s, 16, dict(base, cluster_buff=1), sample_idx=idx, ctx=gpu_ctx)
    red = train.run_basis_train_signal(s, 8, dict(base, cluster_buff=2, kmeans_seed=3), sample_idx=idx, ctx=gpu_ctx)
    _, _, D = kmeans_oracle.kmeans_cityblock(full["B_Mel_sub"].T, 8, seed=3)
    assert red["B_DFT_sub"].shape == (513, 8) and red["B_Mel_sub"].shape == (64, 8) and red["A_DFT_sub"].shape[0] == 8
    # which atoms were kept: every reduced column is one of the 16 trained ones, bit for bit
    keep = np.array([int(np.flatnonzero((full["B_Mel_sub"] == red["B_Mel_sub"][:, [j]]).all(0))[0]) for j in range(8)])
    # ... the one nearest to its cluster's centroid (a two-member cluster has both members equidistant from their median,
    # so WHICH of the two is kept is a matter of rounding -- in MATLAB as here; the distance is what is checked)
    np.testing.assert_allclose(D[keep, np.arange(8)], D.min(0), rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(red["B_DFT_sub"], full["B_DFT_sub"][:, keep])
    np.testing.assert_array_equal(red["A_DFT_sub"], full["A_DFT_sub"][keep])
    np.testing.assert_array_equal(red["A_Mel_sub"], full["A_Mel_sub"][keep])
    with pytest.raises(SnmfError):  # :125-126 would index a scalar in MATLAB
        train.run_basis_train_signal(s, 8, dict(base, cluster_buff=2, train_Exemplar=1), sample_idx=idx, ctx=gpu_ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("mel", [False, True], ids=["run_basis_DNMF", "run_basis_DNMF_Mel"])
def test_dnmf_callers_from_waveforms(gpu_ctx, mel):
    """run_basis_DNMF.m / run_basis_DNMF_Mel.m with the reference's signature (x, d, B, p): equal-length cut,
    y = x + d, three (Mel) feature sets on the device, the 3-solve loop.  Against the oracle chain."""
    import oracle.frontend_oracle as fo
    from se_snmf_nat_amd import train
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"].astype(np.float64)
    x, d = s[:9000], s[9000:19000][::-1].copy()  # two different signals of unequal length
    p = dict(fo.default_params(), cf="kl", sparsity=5, max_iter=12, conv_eps=1e-3, cost_check=1, random_seed=1, R_x=10, R_d=12)
    rs = np.random.RandomState(5)
    F = 64 if mel else 513
    B = rs.rand(F, 22) + 0.05
    ref = fo.run_basis_DNMF(x, d, B, p, mel=mel)
    dev = (train.run_basis_DNMF_Mel if mel else train.run_basis_DNMF)(x, d, B, p, ctx=gpu_ctx)
    assert dev.shape == ref.shape == (F, 22)
    assert np.linalg.norm(dev - ref) / np.linalg.norm(ref) < 1e-4
    np.testing.assert_allclose(np.sqrt((dev.astype(np.float64) ** 2).sum(0)), 1.0, rtol=1e-5)
