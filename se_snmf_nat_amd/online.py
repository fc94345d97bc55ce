"""Host-side mirror of the online separation path (BASELINE config 3).

    g = init_buff(B_Mel_x, B_Mel_d, B_DFT_x, B_DFT_d, p)                   <->  src/init_buff.m:1
    [x_hat_i, d_hat_i, x_tilde, g] = bnmf_sep_event_RT_IS16(y, l, g, p)     <->  src/bnmf_sep_event_RT_IS16.m:1
    NTF_sep_event_RT(path_in, ..., B_DFT_x, B_DFT_d, p)                     <->  src/NTF_sep_event_RT.m:1

`OnlineSeparator` owns the state `g` on the device (snmf_online in include/snmf.h); `process(pcm)` is
the frame loop of the driver for as many hops as `pcm` holds -- framing, STFT, the per-frame solve, the
gain, the noise-dictionary adaptation, inverse STFT and overlap-add all run in libsnmf_hip.so.
`ntf_sep_event_rt` is the file-level call.  Parameter names are the reference's
(settings/initial_setting_SNMF_NAT.m); `default_settings()` returns the shipped values.

Scope = the configuration the reference ships: blk_len_sep = 1, Splice = 0, one channel; B_sep_mode 'DFT'
(shipped) or 'Mel' (MelConv 0/1); supervised or semi-supervised frame solve.  Anything else raises.
MATLAB's global-RNG draws (rand(r,1) per frame solve, rand(R_a, m_a) in init_buff) are explicit
arguments `H0` / `Ad_blk0` (default: numpy RandomState(random_seed) stand-ins).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SnmfOnlineFrame, SnmfOnlineParams
from .api import default_context

__all__ = ["default_settings", "OnlineSeparator", "ntf_sep_event_rt"]


def default_settings():
    """settings/initial_setting_SNMF_NAT.m: the fields the online path reads."""
    fs = 16000
    framelength = int(round(0.040 * fs))
    frameshift = int(round(0.010 * fs))
    fftlength = 2 ** int(np.ceil(np.log2(framelength)))
    n = np.arange(framelength)
    win = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / framelength))  # sqrt(hann(N,'periodic'))
    dcbin = int(np.floor(80 / (fs / fftlength) + 0.5))
    return dict(
        fs=fs, framelength=framelength, frameshift=frameshift, fftlength=fftlength, win_STFT=win, win_ISTFT=win.copy(),
        overlapscale=2 * frameshift / framelength, pow=2, preemph=0.0, DCbin=dcbin, DCbin_back=dcbin,
        nonzerofloor=1e-9, Splice=0, blk_len_sep=1, delay=0 + 1 + int(np.floor(0.040 / 0.010 / 2 + 0.5)),
        B_sep_mode="DFT", MelConv=1, F_order=64, basis_update_N=0, basis_update_E=0,
        adapt_train_N=1, init_N_len=15, R_a=50, m_a=100, overlap_m_a=0.01, Ar_up=1.0,
        blk_sparse=1, P_len_k=60, P_len_l=20, alpha_p=0.4, blk_gap=3,
        ENHANCE_METHOD="MMSE", alpha_eta=0.4, alpha_d=0.6, beta=1.0, beta_max=1000.0,
        cf="kl", sparsity=5, max_iter=100, conv_eps=1e-3, cost_check=1, random_seed=1,
    )


def _beta_div(p):
    cf = p.get("cf", "kl")
    return {"is": 0.0, "kl": 1.0, "ed": 2.0}.get(cf, float(p.get("beta_div", 1.0)))  # src/sparse_nmf.m:99-110


class OnlineSeparator:
    """State `g` of src/init_buff.m + the per-frame function, resident on the GPU."""

    def __init__(self, B_DFT_x, B_DFT_d, p, H0=None, Ad_blk0=None, ctx=None, class_outputs=False, B_Mel_x=None, B_Mel_d=None):
        mode = p.get("B_sep_mode", "DFT")
        if mode not in ("DFT", "Mel") or p.get("Splice", 0) != 0 or p.get("blk_len_sep", 1) != 1:
            raise NotImplementedError("online path: only Splice=0, blk_len_sep=1 (the shipped settings), B_sep_mode 'DFT' or 'Mel'")
        if mode == "Mel" and (B_Mel_x is None or B_Mel_d is None):
            raise ValueError("B_sep_mode='Mel' needs B_Mel_x and B_Mel_d")
        if "cost_check" not in p:
            raise KeyError("Reference to non-existent field 'cost_check'.")  # src/sparse_nmf.m:260
        method = p.get("ENHANCE_METHOD", "MMSE")
        if method not in ("Wiener", "MMSE"):
            raise ValueError("ENHANCE_METHOD must be 'Wiener' or 'MMSE'")
        self._lib = _lib.load()
        self.ctx = ctx or default_context()
        Bx = np.asfortranarray(B_DFT_x, dtype=np.float32)
        Bd = np.asfortranarray(B_DFT_d, dtype=np.float32)
        F = p["fftlength"] // 2 + 1
        if Bx.shape[0] != F or Bd.shape[0] != F:
            raise ValueError(f"dictionaries must have fftlength/2+1 = {F} rows")
        self.F, self.R_x, self.R_d = F, Bx.shape[1], Bd.shape[1]
        r = self.R_x + self.R_d
        rs = np.random.RandomState(int(p.get("random_seed", 1)) or None)
        if H0 is None:
            H0 = rs.random_sample(r)  # stand-in for rand(r,1), src/sparse_nmf.m:133-134
        adapt = int(bool(p.get("adapt_train_N", 0)))
        R_a, m_a = int(p.get("R_a", 1)), int(p.get("m_a", 1))
        if adapt and Ad_blk0 is None:
            Ad_blk0 = rs.random_sample((R_a, m_a))  # stand-in for rand(R_a, m_a), src/init_buff.m:39
        H0 = np.ascontiguousarray(np.asarray(H0, dtype=np.float32).reshape(-1))
        if H0.size != r:
            raise ValueError("H0 must have R_x + R_d entries")
        Ad = None
        if adapt:
            Ad = np.asfortranarray(Ad_blk0, dtype=np.float32)
            if Ad.shape != (R_a, m_a):
                raise ValueError("Ad_blk0 must be R_a x m_a")
        ws = np.ascontiguousarray(p["win_STFT"], dtype=np.float32)
        wi = np.ascontiguousarray(p["win_ISTFT"], dtype=np.float32)
        q = SnmfOnlineParams()
        q.fftlength, q.framelength, q.frameshift = int(p["fftlength"]), int(p["framelength"]), int(p["frameshift"])
        q.dcbin, q.dcbin_back, q.delay = int(p["DCbin"]), int(p.get("DCbin_back", p["DCbin"])), int(p["delay"])
        q.preemph, q.pow, q.nonzerofloor = float(p.get("preemph", 0.0)), float(p.get("pow", 2)), float(p.get("nonzerofloor", 1e-9))
        q.overlapscale = float(p["overlapscale"])
        q.R_x, q.R_d = self.R_x, self.R_d
        q.beta_div, q.sparsity = _beta_div(p), float(p.get("sparsity", 0))
        q.max_iter, q.cost_check, q.conv_eps = int(p.get("max_iter", 100)), int(bool(p["cost_check"])), float(p.get("conv_eps", 0))
        q.enhance_method = 0 if method == "Wiener" else 1
        q.init_N_len = int(p.get("init_N_len", 0))
        q.alpha_eta, q.alpha_d = float(p.get("alpha_eta", 0.4)), float(p.get("alpha_d", 0.6))
        q.beta, q.beta_max = float(p.get("beta", 1.0)), float(p.get("beta_max", 1000.0))
        q.blk_sparse = int(bool(p.get("blk_sparse", 0)))
        q.P_len_k, q.P_len_l, q.blk_gap = int(p.get("P_len_k", 60)), int(p.get("P_len_l", 20)), int(p.get("blk_gap", 3))
        q.alpha_p = float(p.get("alpha_p", 0.4))
        q.adapt_train_N, q.R_a, q.m_a = adapt, R_a, m_a
        q.overlap_m_a, q.Ar_up = float(p.get("overlap_m_a", 0.01)), float(p.get("Ar_up", 1.0))
        q.class_outputs = int(bool(class_outputs))
        q.basis_update_N, q.basis_update_E = int(bool(p.get("basis_update_N", 0))), int(bool(p.get("basis_update_E", 0)))
        self._q = q
        self.class_outputs = bool(class_outputs)
        self.hop, self.delay = q.frameshift, q.delay
        h = C.c_void_p()
        _lib.check(self._lib.snmf_online_create(self.ctx._h, C.byref(q), Bx.ctypes.data, Bd.ctypes.data, H0.ctypes.data,
                                                Ad.ctypes.data if Ad is not None else None, ws.ctypes.data, wi.ctypes.data,
                                                C.byref(h)))
        self._h = h
        self.ctx._plans.add(self)  # destroyed before the context
        self.mel = mode == "Mel"
        if self.mel:
            from .frontend import mel_matrix
            n1 = int(p.get("F_order", 64))
            melmat = np.ascontiguousarray(mel_matrix(p["fs"], n1, p["fftlength"], 1.0, p["fs"] / 2).T, dtype=np.float32)  # init_buff.m:46
            BMx = np.asfortranarray(B_Mel_x, dtype=np.float32)
            BMd = np.asfortranarray(B_Mel_d, dtype=np.float32)
            if BMx.shape != (n1, self.R_x) or BMd.shape != (n1, self.R_d):
                raise ValueError("B_Mel_x / B_Mel_d must be F_order x R_x / R_d")
            self.n1 = n1
            _lib.check(self._lib.snmf_online_set_mel(self._h, n1, int(bool(p.get("MelConv", 1))), melmat.ctypes.data, BMx.ctypes.data,
                                                     BMd.ctypes.data))

    def process(self, pcm, flush=False):
        """Feed PCM (int16 or int16-valued floats).  Returns a dict with the hops the driver writes for the
        frames completed by this call: 'x_tilde' (int16, what fwrite(...,'int16') stores), 'x_tilde_f'
        (float, before rounding) and with class_outputs 'x_hat' / 'd_hat'."""
        x = np.ascontiguousarray(np.asarray(pcm).reshape(-1), dtype=np.float32)
        cap = (x.size // self.hop + self.delay + 3) * self.hop
        of = np.zeros(cap, np.float32)
        o16 = np.zeros(cap, np.int16)
        xh = np.zeros(cap, np.float32) if self.class_outputs else None
        dh = np.zeros(cap, np.float32) if self.class_outputs else None
        n = C.c_int64()
        _lib.check(self._lib.snmf_online_process_f32(
            self._h, x.ctypes.data if x.size else None, x.size, 1 if flush else 0, of.ctypes.data, o16.ctypes.data,
            xh.ctypes.data if xh is not None else None, dh.ctypes.data if dh is not None else None, cap, C.byref(n)))
        out = {"x_tilde": o16[:n.value], "x_tilde_f": of[:n.value]}
        if self.class_outputs:
            out["x_hat"], out["d_hat"] = xh[:n.value], dh[:n.value]
        return out

    def basis(self):
        """Current B_DFT_d (g.B_DFT_d; saved to B_D_u.mat by src/NTF_sep_event_RT.m:138-140)."""
        B = np.zeros((self.F, self.R_d), dtype=np.float32, order="F")
        _lib.check(self._lib.snmf_online_get_basis_f32(self._h, B.ctypes.data, self.F))
        return B.astype(np.float64)

    def mel_basis(self):
        """Current B_Mel_d (Mel mode: the dictionary the adaptation updates, :318)."""
        B = np.zeros((self.n1, self.R_d), dtype=np.float32, order="F")
        _lib.check(self._lib.snmf_online_get_mel_basis_f32(self._h, B.ctypes.data, self.n1))
        return B.astype(np.float64)

    def trace(self):
        """Per-frame diagnostics: list of dicts (n_iter, trig, solved, n_up, adapt_iters, beta, A_x_mag, ...)."""
        n = C.c_int64()
        _lib.check(self._lib.snmf_online_trace(self._h, None, 0, C.byref(n)))
        arr = (SnmfOnlineFrame * max(1, n.value))()
        _lib.check(self._lib.snmf_online_trace(self._h, C.cast(arr, C.c_void_p), n.value, C.byref(n)))
        return [{k: getattr(arr[i], k) for k, _ in SnmfOnlineFrame._fields_} for i in range(n.value)]

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                self._lib.snmf_online_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ntf_sep_event_rt(pcm, B_DFT_x, B_DFT_d, p, H0=None, Ad_blk0=None, ctx=None, chunk=None, B_Mel_x=None, B_Mel_d=None):
    """src/NTF_sep_event_RT.m for one channel with p.NMF_algorithm = 'SNMF': pcm = the int16 samples after the
    wav header.  Returns (denoised int16, denoised float, final B_DFT_d).  `chunk` (samples per process() call)
    only changes how the stream is fed, not the result."""
    sep = OnlineSeparator(B_DFT_x, B_DFT_d, p, H0=H0, Ad_blk0=Ad_blk0, ctx=ctx, B_Mel_x=B_Mel_x, B_Mel_d=B_Mel_d)
    try:
        x = np.asarray(pcm).reshape(-1)
        if chunk is None:
            o = sep.process(x, flush=True)
            i16, f32 = o["x_tilde"], o["x_tilde_f"]
        else:
            parts = [sep.process(x[i:i + chunk]) for i in range(0, len(x), chunk)]
            parts.append(sep.process(x[:0], flush=True))
            i16 = np.concatenate([q["x_tilde"] for q in parts])
            f32 = np.concatenate([q["x_tilde_f"] for q in parts])
        return i16.copy(), f32.astype(np.float64), (sep.mel_basis() if sep.mel else sep.basis())
    finally:
        sep.close()
