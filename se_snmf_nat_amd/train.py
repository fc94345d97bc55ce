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

import ctypes as C

import numpy as np

from . import _lib, frontend
from .api import SnmfError, _make_params, _solver_scalars, default_context


def run_basis_train_signal(s_full, R, p, *, DC_bin=None, sample_idx=None, ctx=None, h0="host"):
    """run_basis_train.m:58-136.  `p`: the reference's settings fields (front-end fields of
    frontend.default_params() plus cf/sparsity/max_iter/conv_eps/cost_check, cluster_buff,
    train_Exemplar).  sample_idx: the exemplar columns (1-based like randsample, :81); default = a
    seeded numpy draw standing in for MATLAB's rng(1); randsample(...).  cluster_buff > 1: the dictionary is trained
    with cluster_buff*R atoms and reduced to R by kmeans.reduce_rank (:118-127; p["kmeans_seed"] seeds its draws).

    ONE call across the C ABI (include/snmf.h: snmf_run_basis_train_audio_f64): TF_mag, TF_DD, TF_Mel, the exemplar columns and
    both solves stay in HBM -- only the audio goes in and the dictionaries (and activations) come out.
    h0: "host" = the rand(r, n) both sparse_nmf calls draw after re-seeding (src/sparse_nmf.m:112-114,:133-134; RandomState
    stand-in) is drawn here and uploaded; "device" = drawn on the device (api.philox_uniform)."""
    p = dict(p)
    cluster_buff = int(p.get("cluster_buff", 1))
    exemplar = bool(p.get("train_Exemplar", 0))
    if cluster_buff > 1 and exemplar:
        # :125-126 index the activation matrices, which :95-96 set to the scalar 0 in exemplar mode: MATLAB stops there too
        raise SnmfError(3, "cluster_buff > 1 needs train_Exemplar = 0 (run_basis_train.m:125-126 index A_*_init, a scalar otherwise)")
    fp = dict(p)
    if DC_bin is not None:
        fp["DCbin"] = int(DC_bin)
    s = np.ascontiguousarray(np.asarray(s_full, dtype=np.float32).reshape(-1))
    sp, _win = frontend._params(fp)
    lib = _lib.load()
    T = int(lib.snmf_stft_num_frames(C.byref(sp), s.size))
    K = 2 * sp.splice + 1
    F, M = K * (sp.fftlength // 2 + 1), int(p["F_order"])
    n_ex = cluster_buff * int(R)
    if T < 1:
        raise SnmfError(3, "the training signal is shorter than one analysis frame")
    if sample_idx is None:
        sample_idx = np.random.RandomState(1).choice(T, size=n_ex, replace=False) + 1  # :80-81 stand-in
    idx0 = np.ascontiguousarray(np.asarray(sample_idx, dtype=np.int64).reshape(-1) - 1)
    if idx0.size != n_ex or idx0.min() < 0 or idx0.max() >= T:
        raise SnmfError(3, "sample_idx must hold cluster_buff*R valid 1-based column indices")
    if exemplar:
        beta, max_iter, conv_eps, cost_check, lam = 1.0, 1, 0.0, 0, 0.0
    else:
        beta, max_iter, conv_eps, cost_check, lam = _solver_scalars(p)
    q = _make_params(F, T, n_ex, beta, max_iter, conv_eps, cost_check, True, 0, lam, None, None)
    seed = int(p.get("random_seed", 1))
    H0 = None
    if h0 == "host" and not exemplar:
        H0 = np.asfortranarray(np.random.RandomState(seed if seed > 0 else None).random_sample((n_ex, T)))
    mel = np.ascontiguousarray(frontend.mel_matrix(p["fs"], M, p["fftlength"], 1.0, p["fs"] / 2).T, dtype=np.float32)
    B_DFT = np.empty((F, n_ex), order="F")
    B_Mel = np.empty((K * M, n_ex), order="F")
    A_DFT = A_Mel = 0  # :95-96
    if not exemplar:
        A_DFT = np.empty((n_ex, T), order="F")
        A_Mel = np.empty((n_ex, T), order="F")
    nit = np.zeros(2, np.int32)
    ptr = lambda a: C.c_void_p(a.ctypes.data) if isinstance(a, np.ndarray) else None
    ctx = ctx or default_context()
    _lib.check(lib.snmf_run_basis_train_audio_f64(
        ctx._h, C.byref(q), C.byref(sp), float(p["alpha_eta"]) if p.get("domain_DD", 0) else -1.0, ptr(mel), M, ptr(s), s.size,
        ptr(idx0), 1 if exemplar else 0, ptr(H0), seed, ptr(B_DFT), ptr(A_DFT), ptr(B_Mel), ptr(A_Mel), ptr(nit)))
    B_DFT = B_DFT / np.sqrt((B_DFT ** 2).sum(0)) + 1e-9  # :113-114
    B_Mel = B_Mel / np.sqrt((B_Mel ** 2).sum(0)) + 1e-9  # :115-116
    if cluster_buff > 1:  # :118-127: keep one basis vector per cluster of the Mel dictionary (host logic, as in the reference)
        from .kmeans import reduce_rank
        B_Mel, B_DFT, A_DFT, A_Mel, _ = reduce_rank(B_Mel, B_DFT, A_DFT, A_Mel, int(R), seed=int(p.get("kmeans_seed", 1)))
    return {"B_DFT_sub": B_DFT, "B_Mel_sub": B_Mel, "A_DFT_sub": A_DFT, "A_Mel_sub": A_Mel}  # :130-134


def save_basis_mat(path, out, v73=True):
    """run_basis_train.m:136: `save(..., 'B_DFT_sub', 'B_Mel_sub', 'A_DFT_sub', 'A_Mel_sub', '-v7.3')` -- an HDF5-based MAT-file
    (se_snmf_nat_amd/mat73.py); v73=False writes MAT-5 like the three .mat files the reference ships (basis/*/R_100.mat, B_D_u.mat).
    MATLAB's `load` reads either."""
    vars_ = {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}
    if v73:
        from .mat73 import save_mat73
        save_mat73(path, vars_)
    else:
        import scipy.io as sio
        sio.savemat(path, vars_, do_compression=True)


def load_basis_mat(path):
    """run_basis_train.m:138 / src/NTF_sep_event_RT.m:28-38: `load` of a dictionary file, MAT-5 or -v7.3 (what the reference's own
    run_basis_train.m:136 writes)."""
    from .mat73 import is_mat73, load_mat73
    if is_mat73(path):
        return load_mat73(path)
    import scipy.io as sio
    m = sio.loadmat(path)
    return {k: v for k, v in m.items() if not k.startswith("__")}


def _dnmf_features(x, d, p, ctx):
    """run_basis_DNMF.m:3-34: equal lengths, y = x + d (waveform sum), three spectrogram feature sets on the GPU
    (returned to the host: the resident entry below forms them in HBM instead)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    n = min(len(x), len(d))  # :5-9
    x, d = x[:n], d[:n]
    y = x + d  # :10
    return tuple(frontend.stft_features(sig, p, ctx=ctx) for sig in (y, x, d))  # :13-34


def _run_basis_dnmf_audio(x, d, B, p, *, ctx, mel, h0, want_a=False):
    """ONE call across the C ABI (snmf_run_basis_dnmf_audio_f64): the waveforms go in, B_hat comes out; features, A_hat and
    all three V matrices stay in HBM."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
    d = np.ascontiguousarray(np.asarray(d, dtype=np.float32).reshape(-1))
    sp, _win = frontend._params(p)
    lib = _lib.load()
    R_x, R_d = int(p["R_x"]), int(p["R_d"])
    r = R_x + R_d
    T = int(lib.snmf_stft_num_frames(C.byref(sp), min(x.size, d.size)))
    K = 2 * sp.splice + 1
    M = int(p["F_order"]) if mel else 0
    F = K * M if mel else K * (sp.fftlength // 2 + 1)
    if T < 1:
        raise SnmfError(3, "the signals are shorter than one analysis frame")
    B = np.asfortranarray(B, dtype=np.float64)
    if B.shape != (F, r):
        raise SnmfError(3, f"B is {B.shape}, expected ({F}, {r})")
    beta, max_iter, conv_eps, cost_check, lam = _solver_scalars(p)
    q = _make_params(F, T, r, beta, max_iter, conv_eps, cost_check, True, 0, lam, None, None)
    seed = int(p.get("random_seed", 1))
    H0 = None
    if h0 == "host":
        H0 = np.asfortranarray(np.random.RandomState(seed if seed > 0 else None).random_sample((r, T)))
    melm = np.ascontiguousarray(frontend.mel_matrix(p["fs"], M, p["fftlength"], 1.0, p["fs"] / 2).T, dtype=np.float32) if mel else None
    B_hat = np.empty((F, r), order="F")
    A_hat = np.empty((r, T), order="F") if want_a else None
    nit = np.zeros(3, np.int32)
    ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
    ctx = ctx or default_context()
    _lib.check(lib.snmf_run_basis_dnmf_audio_f64(ctx._h, C.byref(q), C.byref(sp), R_x, R_d, ptr(x), x.size, ptr(d), d.size, ptr(melm), M,
                                                 ptr(B), F, ptr(H0), seed, ptr(B_hat), F, ptr(A_hat), r, ptr(nit)))
    return (B_hat, A_hat) if want_a else B_hat


def _check_dtype(dtype):
    """`dtype` of the callers below: accepted for callers written against the round-3 signature.  The entry they go through takes
    the waveforms as fp32 (the device front-end's sample type: y = x + d of run_basis_DNMF.m:10 is formed in fp32 on the device,
    INTEGRATION.md) and returns fp64; any other request is an error rather than silently something else."""
    if dtype is not None and np.dtype(dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise SnmfError(1, "dtype must be float64 or float32")


def run_basis_DNMF(x, d, B, p, *, ctx=None, h0="host", dtype=None):
    """B_hat = run_basis_DNMF(x, d, B, p) -- run_basis_DNMF.m:1: clean and noise waveforms, exemplar basis
    B = [B_x, B_d] (F x (R_x+R_d)); p carries the front-end fields, R_x, R_d and the solver fields."""
    _check_dtype(dtype)
    return _run_basis_dnmf_audio(x, d, B, p, ctx=ctx, mel=False, h0=h0)  # :1-55


def run_basis_DNMF_Mel(x, d, B, p, *, ctx=None, h0="host", dtype=None):
    """B_hat = run_basis_DNMF_Mel(x, d, B, p) -- run_basis_DNMF_Mel.m:1: the same loop on the Mel projections of
    the three feature sets (:21-69); B is the Mel exemplar basis (F_order*(2*Splice+1) rows)."""
    _check_dtype(dtype)
    return _run_basis_dnmf_audio(x, d, B, p, ctx=ctx, mel=True, h0=h0)  # :1-95
