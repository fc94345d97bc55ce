"""Host mirror of the basis-training caller, run_basis_train.m:58-136, for ONE event class whose
training signal has already been assembled (the wav shuffling / concatenation of :16-57 is file I/O
and stays with the caller).  Everything numeric runs on the GPU: features (frontend.py), both
sparse_nmf solves (api.py); this module only sequences them like the reference does.

    out = run_basis_train_signal(s_full, R, p)        # B_DFT_sub, B_Mel_sub, A_DFT_sub, A_Mel_sub
    save_basis_mat(path, out)                         # R_<R>.mat with the reference's variable names
    B_hat = run_basis_DNMF(x, d, B, p)                # run_basis_DNMF.m:1      (waveforms in, like the reference)
    B_hat = run_basis_DNMF_Mel(x, d, B, p)            # run_basis_DNMF_Mel.m:1
"""
from __future__ import annotations

import numpy as np

from . import frontend
from .api import SnmfError, run_basis_dnmf, sparse_nmf


def run_basis_train_signal(s_full, R, p, *, DC_bin=None, sample_idx=None, ctx=None):
    """run_basis_train.m:58-136.  `p`: the reference's settings fields (front-end fields of
    frontend.default_params() plus cf/sparsity/max_iter/conv_eps/cost_check, cluster_buff,
    train_Exemplar).  sample_idx: the exemplar columns (1-based like randsample, :81); default = a
    seeded numpy draw standing in for MATLAB's rng(1); randsample(...).  cluster_buff > 1: the dictionary is trained
    with cluster_buff*R atoms and reduced to R by kmeans.reduce_rank (:118-127; p["kmeans_seed"] seeds its draws)."""
    p = dict(p)
    cluster_buff = int(p.get("cluster_buff", 1))
    if cluster_buff > 1 and p.get("train_Exemplar", 0):
        # :125-126 index the activation matrices, which :95-96 set to the scalar 0 in exemplar mode: MATLAB stops there too
        raise SnmfError(3, "cluster_buff > 1 needs train_Exemplar = 0 (run_basis_train.m:125-126 index A_*_init, a scalar otherwise)")
    fp = dict(p)
    if DC_bin is not None:
        fp["DCbin"] = int(DC_bin)
    TF_mag = frontend.stft_features(s_full, fp, ctx=ctx)  # :60-63 (all-zero columns never produced)
    if p.get("domain_DD", 0):
        TF_mag = frontend.tf_dd(TF_mag, p, ctx=ctx)  # :64-67 (src/TF_DD.m)
    TF_Mel = frontend.mel_features(TF_mag, fp, ctx=ctx)  # :70-78
    n_ex = cluster_buff * int(R)
    T = TF_mag.shape[1]
    if sample_idx is None:
        sample_idx = np.random.RandomState(1).choice(T, size=n_ex, replace=False) + 1  # :80-81 stand-in
    sample_idx = np.asarray(sample_idx, dtype=int) - 1
    if sample_idx.size != n_ex or sample_idx.min() < 0 or sample_idx.max() >= T:
        raise SnmfError(3, "sample_idx must hold cluster_buff*R valid 1-based column indices")
    B_DFT = TF_mag[:, sample_idx].astype(np.float64)  # :82
    B_Mel = TF_Mel[:, sample_idx].astype(np.float64)  # :83
    A_DFT = A_Mel = 0
    if not p.get("train_Exemplar", 0):  # :84
        q = {k: p[k] for k in ("cf", "beta", "sparsity", "max_iter", "conv_eps", "cost_check", "random_seed") if k in p}
        q["w_update_ind"] = np.ones(n_ex, bool)  # :85
        q["h_update_ind"] = np.ones(n_ex, bool)  # :86
        q["init_w"] = B_DFT  # :87
        B_DFT, A_DFT, _ = sparse_nmf(TF_mag, q, ctx=ctx)  # :88
        q["init_w"] = B_Mel  # :90
        B_Mel, A_Mel, _ = sparse_nmf(TF_Mel, q, ctx=ctx)  # :91
    B_DFT = B_DFT / np.sqrt((B_DFT ** 2).sum(0)) + 1e-9  # :113-114
    B_Mel = B_Mel / np.sqrt((B_Mel ** 2).sum(0)) + 1e-9  # :115-116
    if cluster_buff > 1:  # :118-127: keep one basis vector per cluster of the Mel dictionary (host logic, as in the reference)
        from .kmeans import reduce_rank
        B_Mel, B_DFT, A_DFT, A_Mel, _ = reduce_rank(B_Mel, B_DFT, A_DFT, A_Mel, int(R), seed=int(p.get("kmeans_seed", 1)))
    return {"B_DFT_sub": B_DFT, "B_Mel_sub": B_Mel, "A_DFT_sub": A_DFT, "A_Mel_sub": A_Mel}  # :130-134


def save_basis_mat(path, out):
    """run_basis_train.m:136 (`save ... -v7.3`); written as MAT-5 like the three .mat files the
    reference ships (basis/*/R_100.mat, B_D_u.mat), which MATLAB's `load` reads the same way."""
    import scipy.io as sio
    sio.savemat(path, {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}, do_compression=True)


def load_basis_mat(path):
    import scipy.io as sio
    m = sio.loadmat(path)
    return {k: v for k, v in m.items() if not k.startswith("__")}


def _dnmf_features(x, d, p, ctx):
    """run_basis_DNMF.m:3-34: equal lengths, y = x + d (waveform sum), three spectrogram feature sets on the GPU."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    n = min(len(x), len(d))  # :5-9
    x, d = x[:n], d[:n]
    y = x + d  # :10
    return tuple(frontend.stft_features(sig, p, ctx=ctx) for sig in (y, x, d))  # :13-34


def run_basis_DNMF(x, d, B, p, *, ctx=None, dtype=np.float32):
    """B_hat = run_basis_DNMF(x, d, B, p) -- run_basis_DNMF.m:1: clean and noise waveforms, exemplar basis
    B = [B_x, B_d] (F x (R_x+R_d)); p carries the front-end fields, R_x, R_d and the solver fields."""
    Y, X, D = _dnmf_features(x, d, p, ctx)
    B_hat, _ = run_basis_dnmf(Y, X, D, B, int(p["R_x"]), int(p["R_d"]), p, ctx=ctx, dtype=dtype)  # :36-55
    return B_hat


def run_basis_DNMF_Mel(x, d, B, p, *, ctx=None, dtype=np.float32):
    """B_hat = run_basis_DNMF_Mel(x, d, B, p) -- run_basis_DNMF_Mel.m:1: the same loop on the Mel projections of
    the three feature sets (:21-69); B is the Mel exemplar basis (F_order*(2*Splice+1) rows)."""
    Y, X, D = _dnmf_features(x, d, p, ctx)
    Ym, Xm, Dm = (frontend.mel_features(M, p, ctx=ctx) for M in (Y, X, D))
    B_hat, _ = run_basis_dnmf(Ym, Xm, Dm, B, int(p["R_x"]), int(p["R_d"]), p, ctx=ctx, dtype=dtype)  # :71-90
    return B_hat
