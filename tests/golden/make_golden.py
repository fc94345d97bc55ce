"""Generates tests/golden/*.npz: inputs + expected outputs of the fp64 oracle restatement of
src/sparse_nmf.m (oracle/sparse_nmf_oracle.py) on small seeded cases, plus fp32 fixtures derived
from the reference's shipped data files (dictionaries basis/*/R_100.mat, B_D_u.mat and the first
frames of wav/M03_423C0213_STR.CH6.wav) used as realistic inputs.

PARITY UNPINNED: the reference has no tests/golden vectors and cannot run here (MATLAB), so these
vectors pin the build's own restatement, not MATLAB output.  Run from the repo root in the build
container (needs /root/reference for the data-derived fixtures):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.sparse_nmf_oracle import sparse_nmf, synth_problem, run_basis_dnmf_solves  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def f32(a):
    """Inputs are rounded to fp32 BEFORE the oracle runs, so the fp32 engine sees them exactly."""
    return np.asarray(a, dtype=np.float32)


def save(name, V, p, shared=(), **extra):
    """shared: names of inputs that live in ref_data.npz instead of this file (V / W0)."""
    V = f32(V)
    p = dict(p, init_w=f32(p["init_w"]), init_h=f32(p["init_h"]))
    w, h, o = sparse_nmf(V.astype(np.float64), {k: (v.astype(np.float64) if k in ("init_w", "init_h") else v)
                                                for k, v in p.items()})
    inputs = dict(V=V, W0=p["init_w"], H0=p["init_h"])
    for k in shared:
        inputs.pop(k)
    d = dict(**inputs, W=w, H=h, div=o["div"], cost=o["cost"], n_iter=o["n_iter"],
             beta={"is": 0.0, "kl": 1.0, "ed": 2.0}.get(p.get("cf", "kl"), p.get("beta", 1.0)),
             sparsity=np.asarray(p["sparsity"], dtype=np.float64), max_iter=p["max_iter"],
             conv_eps=p.get("conv_eps", 0.0),
             w_update_ind=np.asarray(p.get("w_update_ind", np.ones(w.shape[1], bool))),
             h_update_ind=np.asarray(p.get("h_update_ind", np.ones(w.shape[1], bool))), **extra)
    if "W0" in shared and not np.asarray(p.get("w_update_ind", [True])).any():
        d.pop("W")  # W-fixed solves return the (re-normalised) input dictionary
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "n_iter", o["n_iter"], "cost", o["cost"][-1] if len(o["cost"]) else None)


def stft_power(x, frame=640, hop=160, nfft=1024, dcbin=5):
    """|STFT|^2 the way the online path forms it (settings/initial_setting_SNMF_NAT.m:21-37,
    src/bnmf_sep_event_RT_IS16.m:67-78): sqrt-periodic-Hann, zero DC bins, + 1e-9."""
    win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * np.arange(frame) / frame))
    n = 1 + (len(x) - frame) // hop
    out = np.empty((nfft // 2 + 1, n))
    for i in range(n):
        Y = np.fft.rfft(x[i * hop:i * hop + frame] * win, nfft)
        m = np.abs(Y) ** 2
        m[:dcbin] = 0.0
        out[:, i] = m + 1e-9
    return out


def main():
    # 1) scaled-down C1: KL, full update, fixed iteration count
    V, W0, H0 = synth_problem(257, 512, 40)
    save("kl_full_257x512_r40", V, dict(cf="kl", sparsity=5, max_iter=30, conv_eps=0, init_w=W0, init_h=H0, cost_check=1))
    # 2) early stop
    save("kl_stop_257x512_r40", V, dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, init_w=W0, init_h=H0, cost_check=1))
    # 3) beta = 2 and beta = 0, generic 0.5
    V, W0, H0 = synth_problem(129, 300, 24)
    save("ed_full_129x300_r24", V, dict(cf="ed", sparsity=2, max_iter=25, init_w=W0, init_h=H0, cost_check=1))
    save("is_full_129x300_r24", V, dict(cf="is", sparsity=0.05, max_iter=25, init_w=W0, init_h=H0, cost_check=1))
    save("b05_full_129x300_r24", V, dict(cf="beta", beta=0.5, sparsity=0.5, max_iter=25, init_w=W0, init_h=H0, cost_check=1))
    # 4) masks: semi-supervised partial w_update_ind
    V, W0, H0 = synth_problem(64, 200, 48)
    save("kl_semi_64x200_r48", V, dict(cf="kl", sparsity=1, max_iter=20, init_w=W0, init_h=H0, cost_check=1,
                                       w_update_ind=np.arange(48) >= 30))
    if not os.path.isdir(REF):
        print("no /root/reference: data-derived fixtures skipped")
        return
    import scipy.io as sio
    import wave
    Bx = sio.loadmat(f"{REF}/basis/Clean_train_TIMIT_test/TASLP_Splice0-SNMF_p2_DD0/R_100.mat")["B_DFT_sub"]
    Bd = sio.loadmat(f"{REF}/basis/CHiME3_bgn_ch6/TASLP_Splice0-SNMF_p2_DD0/R_100.mat")["B_DFT_sub"]
    Bu = sio.loadmat(f"{REF}/B_D_u.mat")["B_DFT_d"]
    with wave.open(f"{REF}/wav/M03_423C0213_STR.CH6.wav") as wf:
        x = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16).astype(np.float64)
    Y = f32(stft_power(x)[:, 100:164])  # 64 frames of real power spectra, 513 x 64
    B = f32(np.concatenate([Bx, Bd], axis=1))  # 513 x 200 shipped dictionaries
    Bu = f32(Bu[:, :50])
    np.savez_compressed(os.path.join(OUT, "ref_data.npz"), Y=Y, B=B, Bu=Bu)
    rs = np.random.RandomState(1)
    # 5) online H-only, one frame (src/bnmf_sep_event_RT_IS16.m:138-154), shipped W, early stop
    H0 = rs.random_sample((200, 1))
    for j, col in enumerate((0, 17, 40)):
        save(f"online_honly_513x1_r200_f{j}", Y[:, col:col + 1], shared=("V", "W0"), col=col, p=
             dict(cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, init_w=B, init_h=H0, cost_check=1,
                  w_update_ind=np.zeros(200, bool), h_update_ind=np.ones(200, bool)))
    # 6) batched H-only over 64 frames (run_basis_DNMF.m:37-40 shape)
    H0 = rs.random_sample((200, 64))
    save("dnmf_honly_513x64_r200", Y, dict(cf="kl", sparsity=5, max_iter=40, conv_eps=0, init_w=B, init_h=H0,
                                           cost_check=1, w_update_ind=np.zeros(200, bool)), shared=("V", "W0"))
    # 7) noise-dictionary adaptation: W-only, r = 50, 513 x 64 (src/bnmf_sep_event_RT_IS16.m:331-335)
    H0 = rs.random_sample((50, 64)) * 1e6
    save("adapt_wonly_513x64_r50", Y, dict(cf="kl", sparsity=5, max_iter=40, conv_eps=1e-3, init_w=Bu,
                                           init_h=H0, cost_check=1, w_update_ind=np.ones(50, bool),
                                           h_update_ind=np.zeros(50, bool)), shared=("V",))
    # 8) the 3-solve DNMF loop (run_basis_DNMF.m:36-55) with 20+20 shipped bases
    mask = f32(rs.random_sample(Y.shape))
    X = f32(Y * mask + 1e-9)
    D = f32(Y - X + 2e-9)
    Bs = f32(np.concatenate([B[:, :20], B[:, 100:120]], axis=1))
    p = dict(cf="kl", sparsity=5, max_iter=30, conv_eps=1e-3, cost_check=1)
    B_hat, A_hat = run_basis_dnmf_solves(Y.astype(np.float64), X.astype(np.float64), D.astype(np.float64),
                                         Bs.astype(np.float64), 20, 20, p)
    np.savez_compressed(os.path.join(OUT, "dnmf_loop_513x64_r20_20.npz"), mask=mask, B_hat=B_hat, A_hat=A_hat)
    print("dnmf loop ok")


if __name__ == "__main__":
    main()
