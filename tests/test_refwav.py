"""The reference-held pin (SURVEY.md §8c / VERDICT r1 #1): the only outputs of the reference's OWN code in the
reference repository are two recordings processed by MATLAB (filewise_run_IS16.m:6-10 -> src/NTF_sep_event_RT.m:54-139
with the shipped settings and the shipped R_100 dictionaries):

    wav/M03_423C0213_STR.CH6.wav -> wav/M03_423C0213_STR.CH6_out_v3.9_18.wav   (343 hops)
    wav/LM_in.wav                -> wav/LM_in_out_v3.9_18.wav                   (1773 hops)

tests/golden/refwav_pairs.npz holds their int16 samples (tests/golden/make_golden_refwav.py).  The whole online chain
(STFT, per-frame H-only sparse_nmf, blk_sparse, MMSE gain, W-only noise-dictionary adaptation, ISTFT, overlap-add,
the delay+1 flush) is run over the inputs -- by the fp64 oracle on the CPU and by the HIP path on the GPU -- and
compared with what MATLAB wrote.

This is a SOFT pin, and the thresholds say so: MATLAB's legacy generator state (rand('seed',1) per solve,
src/sparse_nmf.m:112-114; the un-re-seeded rand draws of src/init_buff.m:38-39) is not recoverable, and the loop is a
feedback system in which one different stop decision changes every later frame (DESIGN §6c "parity horizon").
Measured when the fixture was made (oracle, NumPy RandomState(1) stand-ins):

    pair   length   best lag   corr(out, ref-out)   gain    SNR of ref-out vs gain*out
    m03    exact    0          0.9967               1.015   21.8 dB      (corr(input, ref-out) = 0.796)
    lm     exact    0          0.9958               1.017   20.7 dB      (corr(input, ref-out) = 0.908)

Counter-experiments: every single-setting deviation from settings/initial_setting_SNMF_NAT.m lowers the
agreement (max_iter 25: 21.6 dB, sparsity 1: 13.2 dB, Wiener: 11.0 dB, no adaptation: 16.5 dB, no block sparsity:
15.2 dB, B_D_u.mat as the start dictionary: 11 dB), so the chain and the shipped settings are what MATLAB ran.

Asserted: output length exact; best lag 0; correlation >= 0.99; SNR >= 20 dB (m03) / >= 19.5 dB (lm).
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THRESH = {"m03": (0.99, 20.0), "lm": (0.99, 19.5)}


def _inputs(key):
    d = np.load(os.path.join(GOLD, "refwav_pairs.npz"))
    B = np.load(os.path.join(GOLD, "ref_data.npz"))["B"].astype(np.float64)  # [R_100 TIMIT | R_100 CHiME3], fp32-rounded
    rs = np.random.RandomState(1)
    H0 = rs.random_sample(200)
    Ad0 = rs.random_sample((50, 100))
    return d[key + "_in"], d[key + "_out"], B[:, :100], B[:, 100:], H0, Ad0


def agreement(out_f, ref_i16):
    """(length equal, best lag in -8..8 samples, correlation, least-squares gain, SNR in dB of ref vs gain*out)."""
    ref = ref_i16.astype(np.float64)
    if len(out_f) != len(ref):
        return False, None, 0.0, 0.0, -np.inf
    o = np.asarray(out_f, dtype=np.float64)

    def xc(L):
        a = o[max(0, L):len(o) + min(0, L)]
        b = ref[max(0, -L):len(ref) + min(0, -L)]
        return float(np.dot(a, b))
    lag = max(range(-8, 9), key=xc)
    corr = float(np.corrcoef(o, ref)[0, 1])
    gain = float(np.dot(o, ref) / np.dot(o, o))
    snr = float(10 * np.log10(np.sum(ref ** 2) / np.sum((ref - gain * o) ** 2)))
    return True, lag, corr, gain, snr


def _check(key, out_f, out_i16, ref):
    ok_len, lag, corr, gain, snr = agreement(out_f, ref)
    print(f"refwav[{key}]: len={len(out_f)} lag={lag} corr={corr:.4f} gain={gain:.4f} snr={snr:.2f} dB")
    cmin, smin = THRESH[key]
    assert ok_len, (len(out_f), len(ref))
    assert len(out_i16) == len(ref)
    assert lag == 0
    assert corr >= cmin, corr
    assert snr >= smin, snr
    assert 0.95 < gain < 1.05, gain


@pytest.mark.parametrize("key", ["m03", "lm"])
def test_oracle_reproduces_the_reference_held_outputs(key):
    from oracle.online_oracle import default_params, ntf_sep_event_rt
    x, ref, Bx, Bd, H0, Ad0 = _inputs(key)
    o16, of, _ = ntf_sep_event_rt(x, Bx, Bd, default_params(), H0, Ad0)
    _check(key, of, o16, ref)


def test_reference_held_outputs_are_not_the_input():
    """The pin is not vacuous: the recordings MATLAB wrote differ from the inputs far more than from the oracle."""
    for key in ("m03", "lm"):
        x, ref, *_ = _inputs(key)
        n = min(len(x), len(ref))
        c = np.corrcoef(x[:n].astype(np.float64), ref[:n].astype(np.float64))[0, 1]
        assert c < 0.92, c


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["m03", "lm"])
def test_device_path_reproduces_the_reference_held_outputs(gpu_ctx, key):
    from se_snmf_nat_amd.online import default_settings, ntf_sep_event_rt
    x, ref, Bx, Bd, H0, Ad0 = _inputs(key)
    o16, of, _ = ntf_sep_event_rt(x, Bx, Bd, default_settings(), H0=H0, Ad_blk0=Ad0, ctx=gpu_ctx)
    _check(key, of, o16, ref)
