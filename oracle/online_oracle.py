"""CPU oracle for the online separation loop around the solver (BASELINE
config 3).  TEST INFRASTRUCTURE ONLY -- the product never imports this module.

PARITY: SOFT PIN -- the reference is MATLAB and cannot run here, but the two recordings it processed itself are
reproduced by this chain to 21.8 dB / 20.7 dB (tests/test_refwav.py); bit-level parity unpinned.
fp64 NumPy restatement, each function citing the reference lines it follows:

  src/init_buff.m:17-42                      state buffers of the SNMF online path
  src/bnmf_sep_event_RT_IS16.m:65-81         per-frame STFT (magnitude^pow with DC bins zeroed + floor, phase)
  src/bnmf_sep_event_RT_IS16.m:124-154       supervised H-only solve of the frame
  src/bnmf_sep_event_RT_IS16.m:158-202       per-class reconstructions and their sums
  src/blk_sparse.m:1-37                      Hoyer block sparsity Q
  src/bnmf_sep_event_RT_IS16.m:220-261       adaptive beta, smoothed noise PSD, Wiener / MMSE gain
  src/bnmf_sep_event_RT_IS16.m:263-347       noise-reference ring buffers, r_up, W-only adaptation solve,
                                             dictionary re-assembly [rem, updated, fixed]
  src/synth_ifft_buff.m:1-32                 per-frame inverse STFT
  src/NTF_sep_event_RT.m:54-135              file-level driver: hop queueing, delay, overlap-add, int16 out

Scope: the configuration the reference ships and runs (settings/initial_setting_SNMF_NAT.m):
blk_len_sep = 1, Splice = 0, one channel; B_sep_mode = 'DFT' (shipped) or 'Mel' (with or without MelConv).
In DFT mode the "Mel" dictionary slots hold the DFT bases (Do_MultiBatch_IS16_20160324.m:199-200), which is what makes the quirks at
:322-328 (row count n1 and the fixed columns taken from B_Mel_d) well defined; they are restated as
written.  MATLAB's global-RNG draws (init_buff.m:38-39 rand for Ad_blk, sparse_nmf.m:133-134 rand(r,1)
for every frame's H0 -- re-seeded identically each frame) are explicit inputs here.
"""
from __future__ import annotations

import numpy as np

from oracle.sparse_nmf_oracle import sparse_nmf

# MATLAB semantics used throughout: max/min skip NaN (np.fmax / np.fmin), x/0 = Inf and 0/0 = NaN without raising.


def default_params():
    """settings/initial_setting_SNMF_NAT.m (the fields the online path reads)."""
    fs = 16000
    framelength = int(round(0.040 * fs))
    frameshift = int(round(0.010 * fs))
    fftlength = 2 ** int(np.ceil(np.log2(framelength)))
    n = np.arange(framelength)
    win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / framelength))
    dcbin = int(np.floor(80 / (fs / fftlength) + 0.5))
    Splice, blk_len_sep = 0, 1
    return dict(
        fs=fs, framelength=framelength, frameshift=frameshift, fftlength=fftlength, win_STFT=win, win_ISTFT=win.copy(),
        overlapscale=2 * frameshift / framelength, pow=2, preemph=0.0, DCbin=dcbin, DCbin_back=dcbin,
        nonzerofloor=1e-9, Splice=Splice, blk_len_sep=blk_len_sep,
        delay=Splice + blk_len_sep + int(np.floor(0.040 / 0.010 / 2 + 0.5)),  # :26  -> 3
        adapt_train_N=1, init_N_len=15, R_a=50, m_a=100, overlap_m_a=0.01, Ar_up=1.0,
        blk_sparse=1, P_len_k=60, P_len_l=20, alpha_p=0.4, blk_gap=3,
        ENHANCE_METHOD="MMSE", alpha_eta=0.4, alpha_d=0.6, beta=1.0, beta_max=1000.0,
        cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, cost_check=1, random_seed=1,
    )


def init_buff(B_DFT_x, B_DFT_d, p, Ad_blk0, mel=None):
    """src/init_buff.m:17-47 (m = 1).  Ad_blk0 stands in for rand(R_a, m_a) at :39.  mel = dict(B_Mel_x, B_Mel_d,
    melmat) for B_sep_mode = 'Mel' (melmat = mel_matrix(fs, F_order, fftlength, 1, fs/2)', :45-47)."""
    n2 = B_DFT_x.shape[0]
    g = dict(
        Xm_tilde=np.zeros(n2), lambda_dav=np.zeros(n2), lambda_Gy=np.zeros(n2),
        r_blk=np.zeros((n2, p["P_len_l"])), Ad_blk=np.array(Ad_blk0, dtype=np.float64),
        lambda_d_blk=np.zeros((n2, p["m_a"])), update_switch=1,
        B_DFT_x=np.array(B_DFT_x, dtype=np.float64), B_DFT_d=np.array(B_DFT_d, dtype=np.float64),
        B_Mel_d=np.array(B_DFT_d, dtype=np.float64),  # DFT mode: Do_MultiBatch_IS16_20160324.m:199-200
    )
    assert g["Ad_blk"].shape == (p["R_a"], p["m_a"])
    if p.get("B_sep_mode", "DFT") == "Mel":
        g["B_Mel_x"] = np.array(mel["B_Mel_x"], dtype=np.float64)
        g["B_Mel_d"] = np.array(mel["B_Mel_d"], dtype=np.float64)
        g["melmat"] = np.array(mel["melmat"], dtype=np.float64)
    return g


def blk_sparse(X, D, r_blk, l, p):
    """src/blk_sparse.m:1-37, literal (1-based k translated to 0-based indices)."""
    K = X.shape[0]
    gapN2 = (p["blk_gap"] - 1) // 2
    snr = X / np.fmax(D, p["nonzerofloor"])  # :10
    with np.errstate(all="ignore"):
        snr = snr / (np.nanmax(snr) if not np.all(np.isnan(snr)) else np.nan)  # :12 (MATLAB max skips NaN)
    r_out = np.concatenate([r_blk[:, 1:p["P_len_l"]], snr[:, None]], axis=1)  # :14
    Q = np.concatenate([np.zeros(p["DCbin"]), 0.1 * np.ones(K - p["DCbin"])])  # :16
    n = p["P_len_l"] * p["P_len_k"]
    k2 = p["P_len_k"] // 2
    if l > p["P_len_l"]:
        for k in range(k2 + p["DCbin"], K - k2 + 1, p["blk_gap"]):  # 1-based k, :20
            b = r_out[k - k2:k + k2, :].reshape(-1)  # rows k-k2+1 .. k+k2 (1-based)
            l1 = b.sum()
            l2 = np.sqrt((b ** 2).sum())
            P_tmp = (np.sqrt(n) - l1 / l2) / (np.sqrt(n) - 1)  # :26
            P_val = p["alpha_p"] * Q[k - 2] + (1 - p["alpha_p"]) * P_tmp  # Q(k-1), :28
            Q[k - gapN2 - 1:k] = P_val  # Q(k-gapN2:k)
            Q[k - 1:k + gapN2] = P_val  # Q(k:k+gapN2)
        Q[:p["P_len_k"] - 1] = Q[p["P_len_k"] + p["DCbin"] - 1]  # :32
    Q[:p["DCbin"]] = 0.0  # :36
    return Q, r_out


def synth_ifft_buff(TF_mag, TF_phase, sz, fftlen, win, preemph, DCbin_back, pw):
    """src/synth_ifft_buff.m:1-32 for ONE frame (column vectors of fftlen/2+1 bins)."""
    mag = np.array(TF_mag, dtype=np.float64)
    mag[:DCbin_back] = 0.0  # :10
    mag = mag ** (1.0 / pw)  # :11
    half = fftlen // 2
    mag_sym = np.concatenate([mag, mag[1:half][::-1]])  # :16
    ph_sym = np.concatenate([TF_phase, -TF_phase[1:half][::-1]])  # :17
    s = np.real(np.fft.ifft(mag_sym * np.exp(1j * ph_sym)))[:sz]  # :18-21
    s = s * win  # :24
    if preemph != 0.0:  # filter(1, [1 -preemph], .)  :26
        out = np.empty_like(s)
        acc = 0.0
        for i in range(sz):
            acc = s[i] + preemph * acc
            out[i] = acc
        s = out
    return s


def frame_stft(y, p):
    """src/bnmf_sep_event_RT_IS16.m:65-81 -> (Ym, Yp)."""
    y = np.asarray(y, dtype=np.float64)
    f = y.copy()
    f[1:] -= p["preemph"] * y[:-1]  # :66
    f = p["win_STFT"] * f  # :67
    pad = np.zeros(p["fftlength"])
    pad[:p["framelength"]] = f
    Y = np.fft.fft(pad)  # :69
    half = p["fftlength"] // 2 + 1
    Yp = np.angle(Y[:half])  # :70
    Ym = np.abs(Y[:half]) ** p["pow"]  # :71
    Ym[:p["DCbin"]] = 0.0  # :74
    Ym = Ym + p["nonzerofloor"]  # :77
    return Ym, Yp


def sep_frame(y, l, g, p, H0):
    """src/bnmf_sep_event_RT_IS16.m for one frame (m = 1, Splice = 0, DFT mode).
    Returns (x_tilde_frame, x_hat_frame, d_hat_frame, trace); g is updated in place."""
    B_x, B_d = g["B_DFT_x"], g["B_DFT_d"]
    n2, R_x = B_x.shape
    R_d = B_d.shape[1]
    flr = p["nonzerofloor"]
    Ym, Yp = frame_stft(y, p)
    mel = p.get("B_sep_mode", "DFT") == "Mel"
    melconv = mel and bool(p.get("MelConv", 1))
    if mel:  # :106-120 feature frequency scale conversion + power normalisation matched to Ym
        melmat = g["melmat"]
        Ym_Mel = melmat @ Ym
        vn = np.sqrt(np.sum(Ym_Mel ** 2))
        tn = np.sqrt(np.sum(Ym ** 2))
        Ym_Mel = (Ym_Mel / vn + 1e-9) * tn
        Y_sep, Bs_x, Bs_d = Ym_Mel, g["B_Mel_x"], g["B_Mel_d"]
    else:
        Y_sep, Bs_x, Bs_d = Ym, B_x, B_d
    # 1) supervised solve (:124-154)
    q = dict(cf=p["cf"], sparsity=p["sparsity"], max_iter=p["max_iter"], conv_eps=p["conv_eps"], cost_check=p["cost_check"],
             init_w=np.concatenate([Bs_x, Bs_d], axis=1), init_h=np.asarray(H0, dtype=np.float64).reshape(-1, 1),
             w_update_ind=np.zeros(R_x + R_d, bool), h_update_ind=np.ones(R_x + R_d, bool))
    if p.get("basis_update_N", 0):  # :125-139 (the third branch, N && E, is unreachable in the reference)
        q["w_update_ind"] = np.concatenate([np.zeros(R_x, bool), np.ones(R_d, bool)])
    elif p.get("basis_update_E", 0):
        q["w_update_ind"] = np.concatenate([np.ones(R_x, bool), np.zeros(R_d, bool)])
    if "beta_div" in p:
        q["beta"] = p["beta_div"]
    _, A, obj = sparse_nmf(Y_sep[:, None], q)
    A = A[:, 0]
    # class sums (:158-202): with any class partition the sums are B_x*A_x and B_d*A_d
    if melconv:  # :165-171,:185-192 and :205-211
        Xs = melmat.T @ (Bs_x @ A[:R_x])
        Ds = melmat.T @ (Bs_d @ A[R_x:])
        Ym_Mel_DFT = melmat.T @ Ym_Mel
    else:  # DFT mode, or coupled dictionaries (Mel activations on the DFT bases)
        Xs = B_x @ A[:R_x]
        Ds = B_d @ A[R_x:]
        Ym_Mel_DFT = Ym
    # block sparsity (:214-218)
    if p["blk_sparse"]:
        Q, g["r_blk"] = blk_sparse(Xs, Ds, g["r_blk"], l, p)
    else:
        Q = np.ones(n2)
    # 3) gain (:221-261)
    if l == 1:
        g["lambda_dav"] = Ym_Mel_DFT.copy()  # :224
    A_d_mag = A[R_x:].sum() / R_d  # :228
    A_x_mag = A[:R_x].sum() / R_x  # :229
    beta = 20 * np.log10(A_d_mag / A_x_mag) * p["beta"]  # :230-231
    if beta < p["beta"]:
        beta = p["beta"]
    elif beta >= p["beta_max"]:
        beta = p["beta_max"]
    g["lambda_dav"] = p["alpha_d"] * g["lambda_dav"] + (1 - p["alpha_d"]) * Ds * beta  # :241
    lambda_d = g["lambda_dav"]
    if p["ENHANCE_METHOD"] == "Wiener":
        G = Xs / (Xs + Ds)  # :245
    else:
        eta = (p["alpha_eta"] * g["Xm_tilde"] + (1 - p["alpha_eta"]) * Xs * Q) / np.fmax(lambda_d, flr)  # :247
        eta = np.fmax(0.0031, eta)  # :251
        G = eta / (eta + 1.0)
    G = np.fmin(G, 1.0)  # :254
    if l <= p["init_N_len"]:  # :256-259
        G = np.zeros(n2) + flr
        A_x_mag = flr
    g["Xm_tilde"] = G * Ym  # :260
    # 4) adaptation (:263-347)
    Q_control = (1 - Q.mean()) * p["Ar_up"]
    trig = bool(p["adapt_train_N"] and (Q_control * A_d_mag > A_x_mag))
    n_up, adapt_iters, solved = 0, 0, False
    if trig:
        if l <= p["init_N_len"]:
            D_ref = Ym.copy()
        else:
            M_ref = 1 - G
            M_ref[:p["DCbin"]] = flr
            D_ref = Ym * M_ref
        g["lambda_Gy"] = D_ref
        g["lambda_d_blk"] = np.concatenate([g["lambda_d_blk"][:, 1:p["m_a"]], D_ref[:, None]], axis=1)  # :282
        g["Ad_blk"] = np.concatenate([g["Ad_blk"][:, 1:p["m_a"]], A[R_x:R_x + p["R_a"], None]], axis=1)  # :285
        r_up = Q_control * g["Ad_blk"].mean(axis=1) > A_x_mag  # :288
        n_up = int(r_up.sum())
        if g["update_switch"] == int(np.floor(p["overlap_m_a"] * p["m_a"])):  # :294
            Ra = p["R_a"]
            Bsrc = g["B_Mel_d"] if mel else B_d  # :298-310 / :322-328
            Vad = (g["melmat"] @ g["lambda_d_blk"]) if mel else g["lambda_d_blk"]  # :299-303
            up = Bsrc[:, :Ra] * r_up[None, :]
            up = up[:, np.any(up != 0, axis=0)]  # :324
            rem = Bsrc[:, :Ra] * (1 - r_up)[None, :]
            rem = rem[:, np.any(rem != 0, axis=0)]  # :326
            fix = g["B_Mel_d"][:, Ra:]  # :309 / :328
            Ad_up = g["Ad_blk"] * r_up[:, None]
            Ad_up = Ad_up[np.any(Ad_up != 0, axis=1), :]  # :292
            if up.shape[1] > 0:
                qa = dict(cf=p["cf"], sparsity=p["sparsity"], max_iter=p["max_iter"], conv_eps=p["conv_eps"],
                          cost_check=p["cost_check"], init_w=up, init_h=Ad_up,
                          w_update_ind=np.ones(up.shape[1], bool), h_update_ind=np.zeros(up.shape[1], bool))
                B_tmp, _, oa = sparse_nmf(Vad, qa)  # :315 / :335
                adapt_iters = oa["n_iter"]
                g["B_Mel_d" if mel else "B_DFT_d"] = np.concatenate([rem, B_tmp, fix], axis=1)  # :318 / :336
                solved = True
            else:
                g["B_Mel_d" if mel else "B_DFT_d"] = np.concatenate([rem, fix], axis=1)
            g["update_switch"] = 1
        else:
            g["update_switch"] += 1
    # inverse STFT (:350-363)
    sy = lambda M: synth_ifft_buff(M, Yp, p["framelength"], p["fftlength"], p["win_ISTFT"], p["preemph"],
                                   p["DCbin_back"], p["pow"]) * p["overlapscale"]
    trace = dict(n_iter=obj["n_iter"], trig=trig, n_up=n_up, solved=solved, adapt_iters=adapt_iters, beta=beta,
                 A_x_mag=A_x_mag, A_d_mag=A_d_mag, Q_control=Q_control, A=A, G=G, Q=Q)
    return sy(g["Xm_tilde"]), sy(Xs), sy(Ds), trace


def _to_int16(x):
    """fwrite(fid, x, 'int16') of doubles: round half away from zero, saturate, NaN -> 0."""
    r = np.floor(np.abs(x) + 0.5) * np.sign(x)
    r = np.where(np.isnan(r), 0.0, r)  # MATLAB integer conversion maps NaN to 0
    return np.clip(r, -32768, 32767).astype(np.int16)


def ntf_sep_event_rt(pcm, B_DFT_x, B_DFT_d, p, H0, Ad_blk0, return_trace=False, class_outputs=False, mel=None):
    """src/NTF_sep_event_RT.m:54-135 for one channel, SNMF algorithm: pcm (int16-valued samples after the
    wav header) -> (denoised int16, float denoised before rounding, final B_DFT_d[, per-frame traces]).
    class_outputs: also overlap-add the event / noise estimates the way :105-119 (commented out in the
    reference) would, appended to the result as (x_hat, d_hat)."""
    pcm = np.asarray(pcm, dtype=np.float64).reshape(-1)
    sz, hop = p["framelength"], p["frameshift"]
    g = init_buff(B_DFT_x, B_DFT_d, p, Ad_blk0, mel)
    y = np.zeros(sz)
    x_tilde = np.zeros(sz)
    xh_buf, dh_buf = np.zeros(sz), np.zeros(sz)
    out, out_x, out_d = [], [], []
    traces = []
    pos, l, cnt_residue = 0, 1, 0
    while True:
        have = pos + hop <= len(pcm)  # fread returned a full hop (:67,:73)
        if cnt_residue > p["delay"]:
            break
        if not have:
            pos = len(pcm)  # a partial hop is consumed by the eof check
            cnt_residue += 1
            y = np.zeros(sz)  # :75
        else:
            y[:sz - hop] = y[hop:].copy()  # :83
            y[sz - hop:] = pcm[pos:pos + hop]  # :84
            pos += hop
        d_frame, e_frame, n_frame, tr = sep_frame(y, l, g, p, H0)
        traces.append(tr)
        if l > p["delay"]:  # :104,:121-124
            x_tilde[:sz - hop] = x_tilde[hop:].copy()
            x_tilde[sz - hop:] = 0.0
            x_tilde = x_tilde + d_frame
            out.append(x_tilde[:hop].copy())
            if class_outputs:  # :105-119
                for buf, fr, dst in ((xh_buf, e_frame, out_x), (dh_buf, n_frame, out_d)):
                    buf[:sz - hop] = buf[hop:].copy()
                    buf[sz - hop:] = 0.0
                    buf += fr
                    dst.append(buf[:hop].copy())
        l += 1
    xf = np.concatenate(out) if out else np.zeros(0)
    res = (_to_int16(xf), xf, g["B_Mel_d"] if p.get("B_sep_mode", "DFT") == "Mel" else g["B_DFT_d"])
    if return_trace:
        res = res + (traces,)
    if class_outputs:
        res = res + (np.concatenate(out_x) if out_x else np.zeros(0), np.concatenate(out_d) if out_d else np.zeros(0))
    return res
