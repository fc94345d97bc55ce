"""Host-side mirror of the reference solver interface, on top of the C ABI (include/snmf.h).

    w, h, objective = sparse_nmf(v, p)          <->  src/sparse_nmf.m:1
    w, h, objective = sparse_nmf_GPU(v, p)      <->  src/sparse_nmf_GPU.m:1
    B_hat, A_hat    = run_basis_dnmf(Y, X, D, B, R_x, R_d, p)   <->  run_basis_DNMF.m:36-55

Same field names, argument meaning, defaults and error behaviour as the MATLAB functions.  All
arithmetic of the solve happens in libsnmf_hip.so on the GPU; this module only does what the
MATLAB wrapper (integration/sparse_nmf.m) does: defaulting (src/sparse_nmf.m:75-164), the random
initial factors (:112-140) and marshalling.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import SnmfError, SnmfParams

__all__ = ["Context", "Plan", "sparse_nmf", "sparse_nmf_GPU", "run_basis_dnmf", "philox_uniform", "dnmf_adapt", "snmf_mdi", "snmf_mdi_Sm", "SnmfError",
           "default_context"]


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class Context:
    """snmf_ctx: one HIP device + stream (replaces the gpuArray state of sparse_nmf_GPU.m:161-166)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.snmf_ctx_create(C.byref(h), int(device)))
        self._h = h
        self.device = int(device)
        self._plans = weakref.WeakSet()  # plans must die before their context (C side keeps a raw pointer)

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._lib.snmf_ctx_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def sync(self):
        _lib.check(self._lib.snmf_ctx_sync(self._h))

    def timing(self, enable):
        _lib.check(self._lib.snmf_ctx_timing(self._h, 1 if enable else 0))

    def timing_get(self, family):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self._lib.snmf_ctx_timing_get(self._h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def xfer_stats(self, reset=False):
        """Host <-> device transfer counters (include/snmf.h: snmf_ctx_xfer_stats)."""
        out = np.zeros(8)
        _lib.check(self._lib.snmf_ctx_xfer_stats(self._h, _ptr(out), 1 if reset else 0))
        keys = ("h2d_bytes", "h2d_wall_s", "h2d_host_copy_s", "h2d_calls", "d2h_bytes", "d2h_wall_s", "d2h_host_copy_s", "d2h_calls")
        return dict(zip(keys, out.tolist()))

    def close(self):
        if getattr(self, "_h", None):
            for pl in list(self._plans):
                pl.close()
            self._lib.snmf_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def _cf_to_beta(p):
    # src/sparse_nmf.m:95-110
    cf = p.get("cf", "kl")
    if cf == "is":
        return 0.0
    if cf == "kl":
        return 1.0
    if cf == "ed":
        return 2.0
    return float(p.get("beta", 1.0))


def _mask(p, key, r):
    m = p.get(key, None)
    if m is None:
        return np.ones(r, np.uint8)  # :142-148
    m = np.ascontiguousarray(np.asarray(m).reshape(-1) != 0, dtype=np.uint8)
    if m.size != r:
        raise SnmfError(3, f"{key} must have r = {r} entries (got {m.size})")
    return m


def _make_params(F, T, r, beta, max_iter, conv_eps, cost_check, floor_v, kind, scalar, w_ind, h_ind):
    sp = SnmfParams()
    sp.F, sp.T, sp.r = int(F), int(T), int(r)
    sp.beta = float(beta)
    sp.max_iter = int(max_iter)
    sp.conv_eps = float(conv_eps)
    sp.cost_check = int(bool(cost_check))
    sp.floor_v = int(bool(floor_v))
    sp.sparsity_kind = int(kind)
    sp.sparsity_scalar = float(scalar)
    sp.w_update_ind = C.c_void_p(w_ind.ctypes.data) if w_ind is not None else None
    sp.h_update_ind = C.c_void_p(h_ind.ctypes.data) if h_ind is not None else None
    return sp


def _sparsity_form(sparsity, r, n, dtype):
    """src/sparse_nmf.m:150-155 -> (kind, scalar, array-or-None)."""
    sp = np.asarray(sparsity, dtype=dtype)
    if sp.size == 1:
        return 0, float(sp.reshape(-1)[0]), None
    if sp.ndim == 1 or (sp.ndim == 2 and sp.shape[1] == 1):
        if sp.size != r:
            raise SnmfError(3, f"sparsity column has {sp.size} rows, h has {r}")
        return 1, 0.0, np.ascontiguousarray(sp.reshape(-1))
    if sp.shape != (r, n):
        raise SnmfError(3, f"sparsity matrix is {sp.shape}, h is ({r}, {n})")
    return 2, 0.0, np.asfortranarray(sp)


def _colmajor(M, dt):
    """M as a column-major array of dtype dt WITHOUT a copy when it already is one (possibly with a leading dimension larger
    than its row count, e.g. the top rows of a taller Fortran array); a copy otherwise."""
    M = np.asarray(M)
    if M.ndim == 2 and M.dtype == dt and M.strides[0] == dt.itemsize and (M.shape[1] <= 1 or M.strides[1] >= M.shape[0] * dt.itemsize):
        return M
    return np.asfortranarray(M, dtype=dt)


def _solve(v, p, *, gpu_variant, ctx, dtype, rng, devices=None):
    p = dict(p or {})
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise SnmfError(1, "dtype must be float64 or float32")
    v = np.asarray(v)
    if v.ndim != 2:
        raise SnmfError(1, "v must be a 2-D matrix")
    m, n = v.shape  # :71-72
    max_iter = int(p.get("max_iter", 100))  # :79-81
    random_seed = p.get("random_seed", 1)  # :83-85
    sparsity = p.get("sparsity", 0)  # :87-89
    conv_eps = float(p.get("conv_eps", 0))  # :91-93
    beta = _cf_to_beta(p)
    if rng is None:  # :112-114 (numpy stand-in for MATLAB's legacy generator)
        rng = np.random.RandomState(int(random_seed) if random_seed and random_seed > 0 else None)

    # :116-131
    if p.get("init_w", None) is None:
        if p.get("r", None) is None:
            raise SnmfError(2, "Number of components or initialization must be given")
        r = int(p["r"])
        w0 = rng.random_sample((m, r))
    else:
        iw = np.asarray(p["init_w"], dtype=np.float64)
        if iw.ndim != 2 or iw.shape[0] != m:
            raise SnmfError(3, f"init_w is {iw.shape}, v has {m} rows")
        ri = iw.shape[1]
        if p.get("r", None) is not None and ri < int(p["r"]):
            r = int(p["r"])
            w0 = np.concatenate([iw, rng.random_sample((m, r - ri))], axis=1)
        else:
            r = ri
            w0 = iw
    # :133-140
    ih = p.get("init_h", None)
    if ih is None:
        h0 = rng.random_sample((r, n))
    elif isinstance(ih, str) and ih == "ones":
        print("sup_nmf: Initalizing H with ones.")  # :135 (unconditional in the reference)
        h0 = np.ones((r, n))
    else:
        h0 = np.asarray(ih, dtype=np.float64)
        if h0.shape != (r, n):
            raise SnmfError(3, f"init_h is {h0.shape}, expected ({r}, {n})")

    w_ind = _mask(p, "w_update_ind", r)
    h_ind = _mask(p, "h_update_ind", r)
    kind, scalar, sarr = _sparsity_form(sparsity, r, n, dt)

    if gpu_variant:
        cost_check = 1  # sparse_nmf_GPU.m:261-277: cost and convergence test unconditional
    else:
        if "cost_check" not in p:  # src/sparse_nmf.m:260
            raise SnmfError(4, "Reference to non-existent field 'cost_check'.")
        cost_check = 1 if p["cost_check"] else 0

    sp = _make_params(m, n, r, beta, max_iter, conv_eps, cost_check, not gpu_variant, kind, scalar, w_ind, h_ind)
    vv = _colmajor(v, dt)
    W = np.asfortranarray(w0, dtype=dt)
    H = np.asfortranarray(h0, dtype=dt)
    div = np.zeros(max(max_iter, 1))
    cost = np.zeros(max(max_iter, 1))
    n_iter = C.c_int32(0)
    lib = _lib.load()
    ldv = vv.strides[1] // dt.itemsize if n > 1 else m
    if devices is not None:
        # the same call over several GPUs: frames sharded over len(devices) ranks, one-shot exchange of the W statistics
        # per iteration (include/snmf.h: snmf_sparse_nmf_multi_*; the MEX shim's opts.devices)
        devs = np.ascontiguousarray(np.asarray(devices, dtype=np.int32).reshape(-1))
        # (the multi-device entry is in/out in W and H: it gets copies, inputs are never modified)
        W, H = W.copy(order="F"), H.copy(order="F")
        fn = lib.snmf_sparse_nmf_multi_f64 if dt == np.float64 else lib.snmf_sparse_nmf_multi_f32
        _lib.check(fn(_ptr(devs), int(devs.size), C.byref(sp), _ptr(vv), ldv, _ptr(W), _ptr(H),
                      _ptr(sarr) if sarr is not None else None, _ptr(div), _ptr(cost), C.byref(n_iter)))
    else:
        # out-of-place entry: init_w / init_h are read where they lie, the results land in fresh arrays (no host copy of init_h)
        ctx = ctx or default_context()
        W0c, H0c = W, H
        W = np.empty((m, r), dtype=dt, order="F")
        H = np.empty((r, n), dtype=dt, order="F")
        fn = lib.snmf_sparse_nmf_oop_f64 if dt == np.float64 else lib.snmf_sparse_nmf_oop_f32
        _lib.check(fn(ctx._h, C.byref(sp), _ptr(vv), ldv, _ptr(W0c), _ptr(H0c), _ptr(sarr) if sarr is not None else None,
                      _ptr(W), _ptr(H), _ptr(div), _ptr(cost), C.byref(n_iter)))
    ni = n_iter.value
    if p.get("display", 0) != 0:  # :162-164 default 0
        _display(beta, div, cost, ni, max_iter, cost_check, conv_eps, gpu_variant)
    if gpu_variant:
        # sparse_nmf_GPU.m:263-264 never fills the vectors: zeros(1, max_iter) are returned
        objective = {"div": np.zeros(max_iter), "cost": np.zeros(max_iter), "n_iter": ni}
    elif cost_check:
        # :279-280 truncate to 1:it on convergence; otherwise full length
        stopped = ni < max_iter
        objective = {"div": div[:ni].copy() if stopped else div[:max_iter].copy(),
                     "cost": cost[:ni].copy() if stopped else cost[:max_iter].copy(), "n_iter": ni}
    else:
        objective = {"div": np.zeros(max_iter), "cost": np.zeros(max_iter), "n_iter": ni}
    return W, H, objective


def _display(beta, div, cost, n_iter, max_iter, cost_check, conv_eps, gpu_variant, out=None):
    """p.display ~= 0: what the reference writes to the console, in its format, reproduced from the objective vectors the
    engine recorded (the whole solve is ONE call across the C boundary, so the lines appear after it instead of during it).
    src/sparse_nmf.m:181-183 (header), :266-270 (per iteration, only inside `if p.cost_check`: the previous line is erased
    with backspaces), :276-278 (convergence), :288-290 -- disp of a SINGLE-quoted string: the backslash-n are printed
    literally, and the line comes after a convergence stop as well.  src/sparse_nmf_GPU.m:266-268,:274: one line per
    iteration, no erasing; its unconditional console chatter (:162,:186,:283, toc) is not reproduced."""
    import sys
    w = (out or sys.stdout).write
    stopped = n_iter < max_iter
    if gpu_variant:
        for it in range(1, n_iter + 1):
            w("iteration %d div = %.3e cost = %.3e\n" % (it, div[it - 1], cost[it - 1]))
        if stopped and conv_eps > 0:
            w("Convergence reached, aborting iteration\n")
        return
    w("Performing sparse NMF with beta-divergence, beta=%.1f\n" % beta)
    if cost_check:
        prev = ""
        for it in range(1, n_iter + 1):
            w("\b" * len(prev))
            prev = "iteration %d div = %.3e cost = %.3e" % (it, div[it - 1], cost[it - 1])
            w(prev)
        if stopped and conv_eps > 0:
            w("Convergence reached, aborting iteration\n")
    w("\\nMax Iteration reached, aborting iteration\\n\n")


def sparse_nmf(v, p=None, *, ctx=None, dtype=np.float64, rng=None, devices=None):
    """[w, h, objective] = sparse_nmf(v, p) -- drop-in for src/sparse_nmf.m on the MI355X.

    `dtype` selects the host-buffer type handed over the C ABI (the device arithmetic is fp32
    MFMA + fp64 objective either way).  `devices`: list of device ordinals -> the frame axis is sharded over
    that many ranks inside this one process (snmf_sparse_nmf_multi_*), same results up to the fp64 summation order
    of the W statistics; the multi-device entry creates a context per rank itself, so `ctx` is not used then."""
    return _solve(v, p, gpu_variant=False, ctx=ctx, dtype=dtype, rng=rng, devices=devices)


def sparse_nmf_GPU(v, p=None, *, ctx=None, dtype=np.float64, rng=None):
    """Drop-in for src/sparse_nmf_GPU.m (its deltas: no V floor, objective vectors left zero,
    cost_check ignored)."""
    return _solve(v, p, gpu_variant=True, ctx=ctx, dtype=dtype, rng=rng)


def philox4x32_10(ctr, key0, key1):
    """Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on arrays of counters:
    ctr = four uint64 arrays holding 32-bit words, key = two 32-bit words.  tests/ check the published known-answer vectors."""
    M32 = np.uint64(0xFFFFFFFF)
    c = [np.asarray(x, dtype=np.uint64) & M32 for x in ctr]
    k0, k1 = np.uint64(key0), np.uint64(key1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M32, p1 & M32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M32, p0 & M32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c


def philox_uniform(seed, r, n):
    """The r x n uniforms the engine draws on the device for an initial H the caller does not supply (include/snmf.h:
    snmf_plan_set_h_random; csrc/snmf_tu_dnmf.hip): Philox-4x32-10 keyed by `seed`, counter = column-major element index // 4,
    value = ((x >> 9) + 0.5) * 2^-23 (exactly representable, strictly inside (0, 1)).  NumPy restatement, so that a host can reproduce the device's draws bit for bit."""
    tot = int(r) * int(n)
    n4 = (tot + 3) // 4
    q = np.arange(n4, dtype=np.uint64)
    c = philox4x32_10([q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.zeros(n4, np.uint64), np.zeros(n4, np.uint64)],
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    x = np.stack(c, axis=1).reshape(-1)[:tot]
    u = ((x >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    return u.reshape((int(n), int(r))).T  # element (k, t) = draw k + r*t


def _solver_scalars(p):
    """(beta, max_iter, conv_eps, cost_check, scalar sparsity) of a settings dict for the callers' entries (run_basis_DNMF.m,
    run_basis_train.m pass the settings struct through: src/sparse_nmf.m:79-110, :260)."""
    if "cost_check" not in p:  # src/sparse_nmf.m:260
        raise SnmfError(4, "Reference to non-existent field 'cost_check'.")
    sp = np.asarray(p.get("sparsity", 0), dtype=np.float64)
    if sp.size != 1:
        raise SnmfError(3, "this caller needs a scalar p.sparsity")
    return _cf_to_beta(p), int(p.get("max_iter", 100)), float(p.get("conv_eps", 0)), 1 if p["cost_check"] else 0, float(sp.reshape(-1)[0])


def run_basis_dnmf(Y, X, D, B, R_x, R_d, p, *, ctx=None, dtype=np.float64, devices=None, resident=None, h0="host"):
    """The 3-solve discriminative re-training loop of run_basis_DNMF.m:36-55 on formed features.

    Y, X, D are the F x T features of mixture / clean / noise (run_basis_DNMF.m:13-34); B is the
    F x (R_x+R_d) exemplar basis.  Returns (B_hat, A_hat).  `devices`: shard the frames of all three solves over
    these GPUs inside this process (BASELINE config 4 behind the reference's own call).
    resident (default: on one device): ONE call of snmf_run_basis_dnmf_* -- Y, X, D uploaded once each, A_hat stays in HBM
    between the solves; False = three separate sparse_nmf calls (the same bits, three host round trips).
    h0: "host" = rand(r, n) of src/sparse_nmf.m:133-134 drawn here (the RandomState stand-in, as sparse_nmf does);
    "device" = drawn on the device (philox_uniform(random_seed, r, n): nothing but the features crosses PCIe); or the
    r x n array itself (what integration/run_basis_DNMF.m passes: MATLAB's own draws)."""
    p = dict(p)
    B = np.asarray(B, dtype=np.float64)
    if resident is None:
        resident = True
    if resident:  # (with a device list: snmf_run_basis_dnmf_multi_*, every rank's shard resident on its device)
        return _run_basis_dnmf_resident(Y, X, D, B, int(R_x), int(R_d), p, ctx=ctx, dtype=dtype, h0=h0, devices=devices)
    p["w_update_ind"] = np.zeros(R_x + R_d, bool)  # :37
    p["h_update_ind"] = np.ones(R_x + R_d, bool)  # :38
    p["init_w"] = B  # :39
    p.pop("init_h", None)
    if isinstance(h0, np.ndarray):
        p["init_h"] = h0
    elif h0 == "device":
        seed = int(p.get("random_seed", 1))
        p["init_h"] = philox_uniform(seed, R_x + R_d, np.asarray(Y).shape[1]).astype(np.float64)
    _, A_hat, _ = sparse_nmf(Y, p, ctx=ctx, dtype=dtype, devices=devices)  # :40
    p["w_update_ind"] = np.ones(R_x, bool)  # :43
    p["h_update_ind"] = np.zeros(R_x, bool)  # :44
    p["init_w"] = B[:, :R_x]  # :45
    p["init_h"] = A_hat[:R_x, :]  # :46
    B_hat_x, _, _ = sparse_nmf(X, p, ctx=ctx, dtype=dtype, devices=devices)  # :47
    p["w_update_ind"] = np.ones(R_d, bool)  # :49
    p["h_update_ind"] = np.zeros(R_d, bool)  # :50
    p["init_w"] = B[:, R_x:R_x + R_d]  # :51
    p["init_h"] = A_hat[R_x:R_x + R_d, :]  # :52
    B_hat_d, _, _ = sparse_nmf(D, p, ctx=ctx, dtype=dtype, devices=devices)  # :53
    return np.concatenate([B_hat_x, B_hat_d], axis=1), A_hat  # :55


def _run_basis_dnmf_resident(Y, X, D, B, R_x, R_d, p, *, ctx, dtype, h0, want_a=True, devices=None):
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise SnmfError(1, "dtype must be float64 or float32")
    Y, X, D = (_colmajor(M, dt) for M in (Y, X, D))
    F, T = Y.shape
    r = R_x + R_d
    if X.shape != (F, T) or D.shape != (F, T):
        raise SnmfError(3, "Y, X and D must have one size")
    if B.shape != (F, r):
        raise SnmfError(3, f"B is {B.shape}, expected ({F}, {r})")
    beta, max_iter, conv_eps, cost_check, lam = _solver_scalars(p)
    sp = _make_params(F, T, r, beta, max_iter, conv_eps, cost_check, True, 0, lam, None, None)
    seed = int(p.get("random_seed", 1))
    H0 = None
    if isinstance(h0, np.ndarray):
        if h0.shape != (r, T):
            raise SnmfError(3, f"init_h is {h0.shape}, expected ({r}, {T})")
        H0 = np.asfortranarray(h0, dtype=dt)
    elif h0 == "host":  # what sparse_nmf draws for solve 1 (:112-114, :133-134; init_w is given, so h is the first draw)
        rs = np.random.RandomState(seed if seed > 0 else None)
        H0 = np.asfortranarray(rs.random_sample((r, T)), dtype=dt)
    Bc = np.asfortranarray(B, dtype=dt)
    B_hat = np.empty((F, r), dtype=dt, order="F")
    A_hat = np.empty((r, T), dtype=dt, order="F") if want_a else None
    nit = np.zeros(3, np.int32)
    lib = _lib.load()
    ldc = lambda M: M.strides[1] // dt.itemsize if M.shape[1] > 1 else M.shape[0]
    tail = (C.byref(sp), R_x, R_d, _ptr(Y), ldc(Y), _ptr(X), ldc(X), _ptr(D), ldc(D), _ptr(Bc), F,
            _ptr(H0) if H0 is not None else None, seed, _ptr(B_hat), F, _ptr(A_hat) if want_a else None, r, _ptr(nit))
    if devices is not None:  # the frames of all three solves sharded over a device list, inside this process
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        fn = lib.snmf_run_basis_dnmf_multi_f64 if dt == np.float64 else lib.snmf_run_basis_dnmf_multi_f32
        _lib.check(fn(_ptr(dv), int(dv.size), *tail))
    else:
        ctx = ctx or default_context()
        fn = lib.snmf_run_basis_dnmf_f64 if dt == np.float64 else lib.snmf_run_basis_dnmf_f32
        _lib.check(fn(ctx._h, *tail))
    return B_hat, A_hat


def dnmf_adapt(Y, D, B, p, *, ctx=None, dtype=np.float64):
    """B_a = DNMF_adapt(Y, D, B, p) -- src/DNMF_adapt.m:1-20: activations of the mixture features Y on the fixed
    dictionary B = [B_x, B_d] (:4-7), then the noise columns re-trained on the noise features D with those
    activations fixed (:16-20).  p carries R_x, R_d and the solver fields."""
    p = dict(p)
    R_x, R_d = int(p["R_x"]), int(p["R_d"])
    B = np.asarray(B, dtype=np.float64)
    p["w_update_ind"] = np.zeros(R_x + R_d, bool)  # :4
    p["h_update_ind"] = np.ones(R_x + R_d, bool)  # :5
    p["init_w"] = B  # :6
    p.pop("init_h", None)
    _, A_hat, _ = sparse_nmf(Y, p, ctx=ctx, dtype=dtype)  # :7
    p["w_update_ind"] = np.ones(R_d, bool)  # :16
    p["h_update_ind"] = np.zeros(R_d, bool)  # :17
    p["init_w"] = B[:, R_x:R_x + R_d]  # :18
    p["init_h"] = A_hat[R_x:R_x + R_d, :]  # :19
    B_a, _, _ = sparse_nmf(D, p, ctx=ctx, dtype=dtype)  # :20
    return B_a


class RcclComm:
    """One rank's RCCL communicator behind the C ABI (snmf_rccl_comm_create): the collective of the process-per-GPU loop without a
    callback into Python.  `unique_id()` on rank 0, ship the 128 bytes to every rank over any channel, then RcclComm(device, id,
    n_ranks, rank) on every rank (collective: ncclCommInitRank)."""

    @staticmethod
    def available():
        return bool(_lib.load().snmf_rccl_available())

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        _lib.check(_lib.load().snmf_rccl_get_unique_id(buf, 128))
        return bytes(buf)

    def __init__(self, device, unique_id, n_ranks, rank):
        self._lib = _lib.load()
        self.handle = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(self._lib.snmf_rccl_comm_create(int(device), buf, int(n_ranks), int(rank), C.byref(self.handle)))
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    def close(self):
        if self.handle:
            self._lib.snmf_rccl_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Plan:
    """snmf_plan: a problem resident in HBM (V, W, H + workspaces).  Arrays may be numpy (host)
    or anything exposing ``data_ptr()``/``dtype`` on the context's device (torch tensors)."""

    def __init__(self, ctx, F, T, r, *, beta=1.0, max_iter=100, conv_eps=0.0, cost_check=True, floor_v=True,
                 sparsity=0.0, w_update_ind=None, h_update_ind=None):
        self.ctx = ctx
        self._lib = _lib.load()
        self.F, self.T, self.r, self.max_iter = int(F), int(T), int(r), int(max_iter)
        self._w_ind = None if w_update_ind is None else np.ascontiguousarray(np.asarray(w_update_ind) != 0, np.uint8)
        self._h_ind = None if h_update_ind is None else np.ascontiguousarray(np.asarray(h_update_ind) != 0, np.uint8)
        self._sarr = None
        if np.ndim(sparsity) == 0:
            kind, scalar = 0, float(sparsity)
        else:
            kind, scalar, self._sarr = _sparsity_form(sparsity, r, T, np.float64)
        sp = _make_params(F, T, r, beta, max_iter, conv_eps, cost_check, floor_v, kind, scalar, self._w_ind,
                          self._h_ind)
        h = C.c_void_p()
        _lib.check(self._lib.snmf_plan_create(ctx._h, C.byref(sp), C.byref(h)))
        self._h = h
        ctx._plans.add(self)
        if self._sarr is not None:
            _lib.check(self._lib.snmf_plan_set_sparsity_f64(self._h, _ptr(self._sarr), 0))

    # -- data ---------------------------------------------------------------------------------
    def _set(self, name, a, rows):
        if hasattr(a, "data_ptr"):  # device tensor, column-major expected: shape (cols, rows) contiguous
            import torch
            if a.dtype == torch.float32:
                ty = "f32"
            elif a.dtype == torch.float64:
                ty = "f64"
            else:
                raise SnmfError(1, "device tensors must be float32 or float64")
            if a.dim() != 2 or not a.is_contiguous() or a.shape[1] != rows:
                raise SnmfError(1, f"device tensor for {name} must be contiguous with shape (cols, {rows}) "
                                   "(= column-major rows x cols)")
            # The tensor was produced on torch's CURRENT stream; the engine packs it on its own stream, and nothing
            # orders the two.  set_* is not a hot path, so wait for the producer here (an input still being written
            # by e.g. `.cuda().T.contiguous()` would otherwise be packed half-finished, silently).
            torch.cuda.current_stream(a.device).synchronize()
            fn = getattr(self._lib, f"snmf_plan_set_{name}_{ty}")
            _lib.check(fn(self._h, C.c_void_p(a.data_ptr()), rows, 1))
            return
        a = np.asarray(a)
        if a.dtype != np.float32:
            a = a.astype(np.float64, copy=False)
        a = np.asfortranarray(a)
        ty = "f32" if a.dtype == np.float32 else "f64"
        fn = getattr(self._lib, f"snmf_plan_set_{name}_{ty}")
        _lib.check(fn(self._h, _ptr(a), a.shape[0], 0))

    def set_v(self, v):
        self._set("v", v, self.F)

    def set_w(self, w):
        self._set("w", w, self.F)

    def set_h(self, h):
        self._set("h", h, self.r)

    def set_h_random(self, seed=1):
        """h = rand(r, n) drawn on the device (philox_uniform(seed, r, T)): nothing crosses PCIe."""
        _lib.check(self._lib.snmf_plan_set_h_random(self._h, int(seed)))

    def set_mask(self, m):
        """Observed (1) / missing (0) mask, binary or soft: turns the plan into an MDI solve (src/snmf_mdi.m)."""
        self._set("mask", np.asarray(m, dtype=np.float64) if not hasattr(m, "data_ptr") else m, self.F)

    def get_v_mdi(self, dtype=np.float64):
        """v_MDI of src/snmf_mdi.m:296-306."""
        dt = np.dtype(dtype)
        out = np.empty((self.F, self.T), dtype=dt, order="F")
        fn = self._lib.snmf_plan_get_v_mdi_f64 if dt == np.float64 else self._lib.snmf_plan_get_v_mdi_f32
        _lib.check(fn(self._h, _ptr(out), self.F, 0))
        return out

    def init(self):
        _lib.check(self._lib.snmf_plan_init(self._h))

    def run(self, n_iters=None):
        done = C.c_int32()
        _lib.check(self._lib.snmf_plan_run(self._h, self.max_iter if n_iters is None else int(n_iters),
                                           C.byref(done)))
        return done.value

    def run_async(self, n_iters):
        _lib.check(self._lib.snmf_plan_run(self._h, int(n_iters), None))

    # -- step API (frame-sharded multi-GPU) ---------------------------------------------------
    def stats_len(self):
        return int(self._lib.snmf_plan_stats_len(self._h))

    def hstep(self):
        _lib.check(self._lib.snmf_plan_hstep(self._h))

    def wstats(self, stats_ptr):
        _lib.check(self._lib.snmf_plan_wstats(self._h, C.c_void_p(stats_ptr)))

    def wapply(self, stats_ptr):
        _lib.check(self._lib.snmf_plan_wapply(self._h, C.c_void_p(stats_ptr)))

    def objstats(self, stats_ptr):
        _lib.check(self._lib.snmf_plan_objstats(self._h, C.c_void_p(stats_ptr)))

    def objapply(self, stats_ptr):
        _lib.check(self._lib.snmf_plan_objapply(self._h, C.c_void_p(stats_ptr)))

    def stopped(self):
        s = C.c_int32()
        _lib.check(self._lib.snmf_plan_stopped(self._h, C.byref(s)))
        return bool(s.value)

    def run_sharded(self, n_iters, stats_ptr, all_reduce=None, *, poll_every=4, finalize=True):
        """snmf_plan_run_sharded: up to n_iters iterations of hstep -> wstats -> all_reduce -> wapply in ONE library call.
        all_reduce(ptr, n_doubles): in-place SUM over the ranks of the n_doubles doubles at device address ptr (None: one rank).
        Returns the iterations run."""
        err = []

        def _cb(ptr, n, _user):
            try:
                all_reduce(int(ptr), int(n))
                return 0
            except BaseException as e:  # (must not propagate through the C frame)
                err.append(e)
                return 1

        cb = _lib.ALLREDUCE_FN(_cb) if all_reduce is not None else _lib.ALLREDUCE_FN()
        done = C.c_int32()
        rc = self._lib.snmf_plan_run_sharded(self._h, int(n_iters), C.c_void_p(stats_ptr), cb, None, int(poll_every),
                                             1 if finalize else 0, C.byref(done))
        if err:
            raise err[0]
        _lib.check(rc)
        return int(done.value)

    def run_sharded_rccl(self, n_iters, stats_ptr, comm, *, poll_every=4, finalize=True):
        """snmf_plan_run_sharded_rccl: the same loop with the library itself issuing ncclAllReduce on the context's stream
        (comm: an RcclComm of this rank).  Returns the iterations run."""
        done = C.c_int32()
        _lib.check(self._lib.snmf_plan_run_sharded_rccl(self._h, int(n_iters), C.c_void_p(stats_ptr), comm.handle, int(poll_every),
                                                        1 if finalize else 0, C.byref(done)))
        return int(done.value)

    # -- results ------------------------------------------------------------------------------
    def get_w(self, dtype=np.float64):
        out = np.empty((self.F, self.r), dtype=dtype, order="F")
        fn = self._lib.snmf_plan_get_w_f64 if out.dtype == np.float64 else self._lib.snmf_plan_get_w_f32
        _lib.check(fn(self._h, _ptr(out), self.F, 0))
        return out

    def get_h(self, dtype=np.float64):
        out = np.empty((self.r, self.T), dtype=dtype, order="F")
        fn = self._lib.snmf_plan_get_h_f64 if out.dtype == np.float64 else self._lib.snmf_plan_get_h_f32
        _lib.check(fn(self._h, _ptr(out), self.r, 0))
        return out

    def get_h_device(self):
        """The activations as a torch CUDA tensor of shape (T, r), float32 (= column-major r x T) on the plan's device: what
        Plan.set_h of another plan takes without a host round trip (dist.run_basis_dnmf_sharded hands solve 1's H to solves 2 / 3)."""
        import torch
        out = torch.empty((self.T, self.r), dtype=torch.float32, device=torch.device("cuda", self.ctx.device))
        torch.cuda.current_stream(out.device).synchronize()
        _lib.check(self._lib.snmf_plan_get_h_f32(self._h, C.c_void_p(out.data_ptr()), self.r, 1))
        self.ctx.sync()
        return out

    def solve_frames(self, v, h0, dtype=np.float64):
        """Online stream: independent H-only solves of consecutive groups of h0.shape[1] (<= 32) columns
        of `v` with the resident dictionary and the same H0 each time
        (src/bnmf_sep_event_RT_IS16.m:138-154, called once per hop by src/NTF_sep_event_RT.m:67-107).
        The plan's T is the capacity in columns.  Returns (H, n_iter, last_cost)."""
        dt = np.dtype(dtype)
        v = np.asfortranarray(v, dtype=dt)
        h0 = np.asfortranarray(h0, dtype=dt)
        tps = h0.shape[1]
        n = v.shape[1] // tps
        if v.shape[0] != self.F or n * tps != v.shape[1] or h0.shape[0] != self.r:
            raise SnmfError(3, "solve_frames: v must be F x (n*tps) and h0 r x tps")
        H = np.empty((self.r, n * tps), dtype=dt, order="F")
        nit = np.zeros(n, np.int32)
        cost = np.zeros(n)
        fn = self._lib.snmf_plan_solve_frames_f64 if dt == np.float64 else self._lib.snmf_plan_solve_frames_f32
        _lib.check(fn(self._h, tps, _ptr(v), v.strides[1] // dt.itemsize if v.shape[1] > 1 else self.F, n, _ptr(h0),
                      _ptr(H), _ptr(nit), _ptr(cost)))
        return H, nit, cost

    def get_objective(self):
        div = np.zeros(max(self.max_iter, 1))
        cost = np.zeros(max(self.max_iter, 1))
        n = C.c_int32()
        _lib.check(self._lib.snmf_plan_get_objective(self._h, _ptr(div), _ptr(cost), C.byref(n)))
        return div[:self.max_iter], cost[:self.max_iter], n.value

    def describe(self):
        buf = C.create_string_buffer(1024)
        _lib.check(self._lib.snmf_plan_describe(self._h, buf, 1024))
        return buf.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):  # a destroyed context has already destroyed its plans
                self._lib.snmf_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _mdi(v, mask, p, *, ctx, dtype, rng):
    """Shared body of snmf_mdi / snmf_mdi_Sm: src/snmf_mdi.m:71-312 (the two files differ only in how the
    mask argument is named and complemented; `.*~Dm` equals `.*(1-Sm)` for a 0/1 mask)."""
    p = dict(p or {})
    dt = np.dtype(dtype)
    v = np.asarray(v)
    mask = np.asarray(mask, dtype=np.float64)
    if v.ndim != 2 or mask.shape != v.shape:
        raise SnmfError(3, "v must be 2-D and the mask must have its size")
    m, n = v.shape
    max_iter = int(p.get("max_iter", 100))  # :83-85
    seed = p.get("random_seed", 1)
    if "sparsity" not in p:  # :87-89 (the default is installed only when p.sparsity is ABSENT)
        p.setdefault("sparsity_mdi", 0)
    if "conv_eps" not in p:  # :91-93
        p.setdefault("conv_eps_mdi", 0)
    for fld in ("sparsity_mdi", "conv_eps_mdi", "cost_check"):
        if fld not in p:
            raise KeyError(f"Reference to non-existent field '{fld}'.")
    beta = _cf_to_beta(p)
    if rng is None:
        rng = np.random.RandomState(int(seed) if seed and seed > 0 else None)  # stand-in for rand('seed',s)
    if p.get("init_w") is None:  # :116-131
        if "r" not in p:
            raise SnmfError(2, "Number of components or initialization must be given")
        w0 = rng.random_sample((m, int(p["r"])))
    else:
        w0 = np.asarray(p["init_w"], dtype=np.float64)
        if p.get("r") is not None and w0.shape[1] < int(p["r"]):
            w0 = np.concatenate([w0, rng.random_sample((m, int(p["r"]) - w0.shape[1]))], axis=1)
    r = w0.shape[1]
    init_h = p.get("init_h")  # :133-140
    if init_h is None:
        h0 = rng.random_sample((r, n))
    elif isinstance(init_h, str) and init_h == "ones":
        h0 = np.ones((r, n))
    else:
        h0 = np.asarray(init_h, dtype=np.float64)
        if h0.shape != (r, n):
            raise SnmfError(3, "init_h must be r x n")
    w_ind, h_ind = _mask(p, "w_update_ind", r), _mask(p, "h_update_ind", r)
    plan = Plan(ctx or default_context(), m, n, r, beta=beta, max_iter=max_iter, conv_eps=float(p["conv_eps_mdi"]),
                cost_check=bool(p["cost_check"]), floor_v=True, sparsity=p["sparsity_mdi"], w_update_ind=w_ind,
                h_update_ind=h_ind)
    try:
        plan.set_mask(mask)
        plan.set_v(v.astype(dt, copy=False) if v.dtype != dt else v)
        plan.set_w(w0)
        plan.set_h(h0)
        plan.init()
        n_it = plan.run()
        v_mdi = plan.get_v_mdi(dtype=dt)
        h = plan.get_h(dtype=dt)
        div, cost, nn = plan.get_objective()
        stopped = plan.stopped()
    finally:
        plan.close()
    k = nn if stopped else max_iter  # :286-287 truncation on convergence only
    return v_mdi, h, {"div": div[:k], "cost": cost[:k], "n_iter": n_it}


def snmf_mdi(v, Dm, p=None, *, ctx=None, dtype=np.float64, rng=None):
    """[v_MDI, h, objective] = snmf_mdi(v, Dm, p)  -- src/snmf_mdi.m:1 (Dm: 1 = observed, 0 = missing)."""
    return _mdi(v, np.asarray(Dm) != 0, p, ctx=ctx, dtype=dtype, rng=rng)


def snmf_mdi_Sm(v, Sm, p=None, *, ctx=None, dtype=np.float64, rng=None):
    """[v_MDI, h, objective] = snmf_mdi_Sm(v, Sm, p)  -- src/snmf_mdi_Sm.m:1 (soft mask in [0,1])."""
    return _mdi(v, Sm, p, ctx=ctx, dtype=dtype, rng=rng)
