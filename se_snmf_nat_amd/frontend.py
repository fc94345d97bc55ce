"""Host mirror of the reference's feature front-end, on top of the C ABI (include/snmf.h).

    stft_features(s, p)       <->  src/stft_fft.m + run_basis_train.m:60-63  (TF_mag, on the GPU)
    mel_features(TF_mag, p)   <->  run_basis_train.m:70-78                   (Mel projection, on the GPU)
    tf_dd(TF_mag, p)          <->  src/TF_DD.m (run_basis_train.m:64-67)     (recursive average along frames, on the GPU)
    mel_matrix(...)           <->  src/mel_matrix.m   (a parameter TABLE, built on the host like the window)
    default_params()          <->  settings/initial_setting_SNMF_NAT.m:17,21-37,53,88-90

`p` uses the reference's field names (framelength, frameshift, fftlength, DCbin, win_STFT, preemph,
pow, nonzerofloor, Splice, fs, F_order).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SnmfError, SnmfStftParams
from .api import default_context


def default_params():
    fs = 16000
    framelength = int(round(0.040 * fs))
    frameshift = int(round(0.010 * fs))
    fftlength = 2 ** int(np.ceil(np.log2(framelength)))
    n = np.arange(framelength)
    return dict(fs=fs, framelength=framelength, frameshift=frameshift, fftlength=fftlength,
                win_STFT=np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / framelength)),  # sqrt(hann(N,'periodic'))
                preemph=0.0, DCbin=int(np.floor(80 / (fs / fftlength) + 0.5)), pow=2, nonzerofloor=1e-9, Splice=0,
                F_order=64)


def _params(p):
    win = np.ascontiguousarray(np.asarray(p["win_STFT"], dtype=np.float64).reshape(-1))
    if win.size != int(p["framelength"]):
        raise SnmfError(3, "win_STFT must have framelength entries")
    sp = SnmfStftParams()
    sp.framelength, sp.frameshift, sp.fftlength = int(p["framelength"]), int(p["frameshift"]), int(p["fftlength"])
    sp.dcbin, sp.splice = int(p["DCbin"]), int(p.get("Splice", 0))
    sp.preemph, sp.pow, sp.nonzerofloor = float(p.get("preemph", 0.0)), float(p.get("pow", 2)), float(p.get("nonzerofloor", 1e-9))
    sp.window = C.c_void_p(win.ctypes.data)
    return sp, win


def num_frames(n_samples, p):
    sp, _win = _params(p)
    return int(_lib.load().snmf_stft_num_frames(C.byref(sp), int(n_samples)))


def stft_features(s, p, *, ctx=None):
    """TF_mag = (|STFT(s)|.^pow + nonzerofloor) with splicing, F x n_frames float32 (Fortran order)."""
    ctx = ctx or default_context()
    s = np.ascontiguousarray(np.asarray(s, dtype=np.float32).reshape(-1))
    sp, _win = _params(p)
    lib = _lib.load()
    nfr = int(lib.snmf_stft_num_frames(C.byref(sp), s.size))
    F = (2 * sp.splice + 1) * (sp.fftlength // 2 + 1)
    out = np.zeros((F, max(nfr, 0)), dtype=np.float32, order="F")
    n_out = C.c_int32()
    _lib.check(lib.snmf_stft_features_f32(ctx._h, C.byref(sp), C.c_void_p(s.ctypes.data), s.size, 0,
                                          C.c_void_p(out.ctypes.data), F, 0, C.byref(n_out)))
    return out


def set_plan_v_from_audio(plan, s, p):
    """V of `plan` := features of the audio `s`, produced in HBM (only the samples cross PCIe)."""
    s = np.ascontiguousarray(np.asarray(s, dtype=np.float32).reshape(-1))
    sp, _win = _params(p)
    _lib.check(_lib.load().snmf_plan_set_v_from_audio_f32(plan._h, C.byref(sp), C.c_void_p(s.ctypes.data), s.size, 0))


def mel_matrix(fs, NbCh, Nfft, warp=1.0, fhigh=None):
    """src/mel_matrix.m:16-38 -> (Nfft/2+1) x NbCh triangular weights (host-side table)."""
    if fhigh is None:
        fhigh = fs / 2
    low = 2595 * np.log10(1 + 64 / 700)
    nyq = 2595 * np.log10(1 + fhigh / 700)
    rnd = lambda x: np.floor(np.abs(x) + 0.5) * np.sign(x)
    start_mel = low + np.arange(NbCh) / (NbCh + 1) * (nyq - low)
    f_cen = warp * 700 * (10 ** (start_mel / 2595) - 1)
    sb = (rnd(Nfft / fs * f_cen) + 1).astype(int)
    end_mel = low + np.arange(2, NbCh + 2) / (NbCh + 1) * (nyq - low)
    eb = (rnd(warp * Nfft / fs * 700 * (10 ** (end_mel / 2595) - 1)) + 1).astype(int)
    tot = eb - sb + 1
    lo = np.concatenate([sb[1:NbCh], [eb[NbCh - 2]]]) - sb + 1
    hi = tot - lo + 1
    M = np.zeros((max(int(np.ceil(warp * Nfft / 2 + 1)), int(eb.max())), NbCh))
    for k in range(NbCh):
        M[sb[k] - 1:sb[k] - 1 + lo[k], k] = np.arange(1, lo[k] + 1) / lo[k]
        M[eb[k] - hi[k]:eb[k], k] = np.arange(hi[k], 0, -1) / hi[k]
    return M[:Nfft // 2 + 1, :]


def mel_features(TF_mag, p, *, ctx=None):
    """run_basis_train.m:70-78 on the GPU: TF_Mel (F_order*(2*Splice+1) x T)."""
    ctx = ctx or default_context()
    n = int(p["fftlength"]) // 2 + 1
    K = 2 * int(p.get("Splice", 0)) + 1
    M = int(p["F_order"])
    mel = np.ascontiguousarray(mel_matrix(p["fs"], M, p["fftlength"], 1.0, p["fs"] / 2).T, dtype=np.float32)
    V = np.asfortranarray(TF_mag, dtype=np.float32)
    if V.shape[0] != K * n:
        raise SnmfError(3, f"TF_mag has {V.shape[0]} rows, expected {K * n}")
    T = V.shape[1]
    out = np.zeros((K * M, T), dtype=np.float32, order="F")
    _lib.check(_lib.load().snmf_mel_features_f32(ctx._h, C.c_void_p(mel.ctypes.data), M, n, K, C.c_void_p(V.ctypes.data),
                                                  K * n, T, C.c_void_p(out.ctypes.data), K * M, 0))
    return out


def tf_dd(X, p, *, ctx=None):
    """[X_DD] = TF_DD(X, p), src/TF_DD.m:1-9: X_DD(:,l) = p.alpha_eta * X_DD(:,l-1) + (1 - p.alpha_eta) * X(:,l), on the GPU."""
    ctx = ctx or default_context()
    if "alpha_eta" not in p:
        raise SnmfError(4, "Reference to non-existent field 'alpha_eta'.")
    X = np.asfortranarray(X, dtype=np.float32)
    if X.ndim != 2:
        raise SnmfError(3, "TF_DD: X must be a matrix")
    F, T = X.shape
    out = np.zeros((F, T), dtype=np.float32, order="F")
    if T == 0:
        return out
    _lib.check(_lib.load().snmf_tf_dd_f32(ctx._h, float(p["alpha_eta"]), F, T, C.c_void_p(X.ctypes.data), F,
                                           C.c_void_p(out.ctypes.data), F, 0))
    return out
