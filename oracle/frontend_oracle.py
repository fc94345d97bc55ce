"""CPU oracle for the spectrogram front-end.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (the reference is MATLAB and cannot run here; see oracle/sparse_nmf_oracle.py).
fp64 NumPy restatements, each citing the reference lines it follows:
  src/stft_fft.m:15-37          framing, pre-emphasis, window, zero-padded FFT, magnitude, DC-bin value
  run_basis_train.m:60-63       drop all-zero columns, splice, .^pow + nonzerofloor
  src/frame_splice.m:1-24       context splicing
  src/mel_matrix.m:16-38        triangular Mel filterbank
  run_basis_train.m:70-78       Mel projection of the (spliced) DFT features
  settings/initial_setting_SNMF_NAT.m:21-37,53,88-90   the shipped parameter values
  run_basis_DNMF.m:1-57, run_basis_DNMF_Mel.m:1-95     the discriminative re-training callers (waveforms in)
Only tests/, __graft_entry__.smoke() and bench scripts' CPU legs may import this module.
"""
from __future__ import annotations

import numpy as np


def default_params():
    """settings/initial_setting_SNMF_NAT.m:17,21-37,53,88-90."""
    fs = 16000
    framelength = int(round(0.040 * fs))  # :29  640
    frameshift = int(round(0.010 * fs))  # :30  160
    fftlength = 2 ** int(np.ceil(np.log2(framelength)))  # :33  1024
    n = np.arange(framelength)
    win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / framelength))  # :37 sqrt(hann(N,'periodic'))
    dcbin = int(np.floor(80 / (fs / fftlength) + 0.5))  # :89-90  5
    return dict(fs=fs, framelength=framelength, frameshift=frameshift, fftlength=fftlength, win_STFT=win,
                preemph=0.0, DCbin=dcbin, pow=2, nonzerofloor=1e-9, Splice=0, F_order=64)


def stft_fft(s, sz, shift, fftlen, DCbin, win, preemph):
    """src/stft_fft.m:15-37.  Returns S_mag (fftlen/2+1 x frame_num) including the trailing columns
    the loop never fills (they stay zero, :18,:21)."""
    s = np.asarray(s, dtype=np.float64).reshape(-1)
    L = len(s)
    frame_num = L // shift  # :17
    half = fftlen // 2 + 1
    S_mag = np.zeros((half, frame_num))  # :18
    size_crnt = 1  # :15 (1-based)
    i = 0
    while size_crnt < L - fftlen:  # :21
        x = s[size_crnt - 1:size_crnt - 1 + sz]  # :22
        y = x.copy()
        y[1:] -= preemph * x[:-1]  # filter([1 -preemph], 1, .) with zero initial state
        y = win * y  # :23
        pad = np.zeros(fftlen)
        pad[:sz] = y  # :25
        S = np.fft.fft(pad)  # :26
        m = np.abs(S[:half])  # :27
        m[:DCbin] = 0.0 + 0.000001  # :31
        S_mag[:, i] = m  # :33
        size_crnt += shift  # :35
        i += 1
    return S_mag


def frame_splice(Feat, Splice):
    """src/frame_splice.m:4-23."""
    K, T = Feat.shape
    out = np.zeros(((2 * Splice + 1) * K, T))
    for t in range(T):  # 0-based t
        for sft in range(Splice + 1):
            hi = slice(K * (Splice + sft), K * (Splice + sft + 1))
            lo = slice(K * (Splice - sft), K * (Splice - sft + 1))
            if (t + 1 - sft) < 1:  # :11
                out[hi, t] = Feat[:, t + sft]
                out[lo, t] = 0.0
            elif (t + 1 + sft) > T:  # :14
                out[hi, t] = 0.0
                out[lo, t] = Feat[:, t - sft]
            else:
                out[hi, t] = Feat[:, t + sft]
                out[lo, t] = Feat[:, t - sft]
    return out


def mel_matrix(fs, NbCh, Nfft, warp=1.0, fhigh=None):
    """src/mel_matrix.m:16-38 -> M (Nfft/2+1 x NbCh)."""
    if fhigh is None:
        fhigh = fs / 2
    LowMel = 2595 * np.log10(1 + 64 / 700)
    NyqMel = 2595 * np.log10(1 + fhigh / 700)
    mround = lambda x: np.floor(np.abs(x) + 0.5) * np.sign(x)  # MATLAB round (half away from zero)
    StartMel = LowMel + np.arange(NbCh) / (NbCh + 1) * (NyqMel - LowMel)
    fCen = warp * 700 * (10 ** (StartMel / 2595) - 1)
    StartBin = (mround(Nfft / fs * fCen) + 1).astype(int)
    EndMel = LowMel + np.arange(2, NbCh + 2) / (NbCh + 1) * (NyqMel - LowMel)
    EndBin = (mround(warp * Nfft / fs * 700 * (10 ** (EndMel / 2595) - 1)) + 1).astype(int)
    TotLen = EndBin - StartBin + 1
    LowLen = np.concatenate([StartBin[1:NbCh], [EndBin[NbCh - 2]]]) - StartBin + 1
    HiLen = TotLen - LowLen + 1
    rows = int(np.ceil(warp * Nfft / 2 + 1))
    M = np.zeros((max(rows, EndBin.max()), NbCh))
    for k in range(NbCh):
        sb, eb = StartBin[k], EndBin[k]  # 1-based
        M[sb - 1:sb - 1 + LowLen[k], k] = np.arange(1, LowLen[k] + 1) / LowLen[k]  # :33
        M[eb - HiLen[k]:eb, k] = np.arange(HiLen[k], 0, -1) / HiLen[k]  # :34
    return M[:Nfft // 2 + 1, :]  # :38


def dft_features(s, p):
    """run_basis_train.m:60-63 (and run_basis_DNMF.m:13-16): TF_mag ready for sparse_nmf."""
    S = stft_fft(s, p["framelength"], p["frameshift"], p["fftlength"], p["DCbin"], p["win_STFT"], p["preemph"])
    S = S[:, np.any(S != 0, axis=0)]  # :61
    S = frame_splice(S, p["Splice"])  # :62
    return S ** p["pow"] + p["nonzerofloor"]  # :63


def mel_features(TF_mag, p):
    """run_basis_train.m:70-78."""
    n = p["fftlength"] // 2 + 1
    melmat = mel_matrix(p["fs"], p["F_order"], p["fftlength"], 1.0, p["fs"] / 2).T  # :72
    K = 2 * p["Splice"] + 1
    out = np.zeros((p["F_order"] * K, TF_mag.shape[1]))
    for k in range(K):
        out[k * p["F_order"]:(k + 1) * p["F_order"], :] = melmat @ TF_mag[k * n:(k + 1) * n, :]  # :76-77
    return out


def tf_dd(X, p):
    """src/TF_DD.m:1-9, fp64, column by column exactly as written."""
    X = np.asarray(X, dtype=np.float64)
    X_DD = X.copy()  # :5
    a = float(p["alpha_eta"])
    for l in range(1, X.shape[1]):  # :6  (l = 2:L)
        X_DD[:, l] = a * X_DD[:, l - 1] + (1 - a) * X[:, l]  # :7
    return X_DD


def run_basis_train_signal(s_full, R, p, sample_idx):
    """run_basis_train.m:58-116 for one event class, cluster_buff = 1, on an assembled signal.
    sample_idx: 1-based exemplar columns (:81).  Uses the solver oracle for :88,:91."""
    from oracle.sparse_nmf_oracle import sparse_nmf
    TF_mag = dft_features(s_full, p)
    if p.get("domain_DD", 0):
        TF_mag = tf_dd(TF_mag, p)  # :64-67
    TF_Mel = mel_features(TF_mag, p)
    idx = np.asarray(sample_idx, dtype=int) - 1
    q = {k: p[k] for k in ("cf", "beta", "sparsity", "max_iter", "conv_eps", "cost_check") if k in p}
    q["init_w"] = TF_mag[:, idx]
    B_DFT, A_DFT, _ = sparse_nmf(TF_mag, q)
    q["init_w"] = TF_Mel[:, idx]
    B_Mel, A_Mel, _ = sparse_nmf(TF_Mel, q)
    B_DFT = B_DFT / np.sqrt((B_DFT ** 2).sum(0)) + 1e-9  # :113-114
    B_Mel = B_Mel / np.sqrt((B_Mel ** 2).sum(0)) + 1e-9  # :115-116
    return {"B_DFT_sub": B_DFT, "B_Mel_sub": B_Mel, "A_DFT_sub": A_DFT, "A_Mel_sub": A_Mel}


def run_basis_DNMF(x, d, B, p, mel=False):
    """run_basis_DNMF.m:1-57 / run_basis_DNMF_Mel.m:1-95 (mel=True).  The rand(r,n) of the first solve is the
    solver oracle's RandomState(p.random_seed) stand-in (the device mirror draws the same)."""
    from oracle.sparse_nmf_oracle import run_basis_dnmf_solves
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    d = np.asarray(d, dtype=np.float64).reshape(-1)
    n = min(len(x), len(d))
    x, d = x[:n], d[:n]
    feats = [dft_features(sig, p) for sig in (x + d, x, d)]
    if mel:
        feats = [mel_features(M, p) for M in feats]
    Y, X, D = feats
    q = {k: p[k] for k in ("cf", "beta", "sparsity", "max_iter", "conv_eps", "cost_check", "random_seed") if k in p}
    return run_basis_dnmf_solves(Y, X, D, np.asarray(B, dtype=np.float64), int(p["R_x"]), int(p["R_d"]), q)[0]
