"""CPU oracle for the sparse-NMF multiplicative-update path.  TEST INFRASTRUCTURE ONLY.

PARITY: SOFT PIN.  The reference (lordet01/SE_SNMF_NAT) is 100 % MATLAB, ships no tests, no golden vectors and no
known-answer fixtures for this path, and neither MATLAB nor Octave exists in the build container or on the GPU box, so
the reference itself cannot be run.  What the reference DOES hold are two recordings its own code processed
(wav/*_out_v3.9_18.wav); tests/test_refwav.py runs the whole online chain -- which calls this solver once per frame
(H-only, KL) and once per adaptation (W-only, KL) -- over the matching inputs and reproduces MATLAB's output to
corr 0.997 / 21.8 dB and 0.996 / 20.7 dB (a soft pin: MATLAB's legacy rand stream is not
recoverable).  Bit-level / 1e-4 parity of a single call, the full-update mode and beta != 1 remain UNPINNED against
MATLAB.  This file is a line-by-line fp64 NumPy restatement of ``src/sparse_nmf.m`` (every block below cites
the lines it follows); it is cross-checked by (i) an independently written loop-form restatement
(``oracle/sparse_nmf_loops.py``), (ii) scikit-learn's ``_beta_divergence`` for the four divergence
formulas, (iii) the algorithm's invariants (monotone cost, unit-norm W, fixed point) and (iv) the
reference's shipped dictionaries (``basis/*/R_100.mat``, ``B_D_u.mat``) used as realistic inputs.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``se_snmf_nat_amd``) never does and has no CPU fallback.

Reference files followed (paths relative to the reference root):
  src/sparse_nmf.m:71-292      parameter handling, init scaling, H/W updates, objective, stop
  src/sparse_nmf_GPU.m         only for the delta list (``gpu_variant=True``)
  run_basis_DNMF.m:36-55       3-solve discriminative loop (``run_basis_dnmf_solves``)
"""
from __future__ import annotations

import numpy as np

FLR = 1e-9  # src/sparse_nmf.m:166


class OracleError(Exception):
    """Mirrors MATLAB `error(...)` on the path (src/sparse_nmf.m:118 and implicit size errors)."""


def _beta_from_cf(p):
    # src/sparse_nmf.m:95-110
    cf = p.get("cf", "kl")
    if cf == "is":
        return 0.0
    if cf == "kl":
        return 1.0
    if cf == "ed":
        return 2.0
    return float(p.get("beta", 1.0))


def divergence(v, lam, beta):
    """src/sparse_nmf.m:248-258 (v already floored, lam already clamped)."""
    if beta == 1:
        return float(np.sum(v * np.log(v / lam) - v + lam))
    if beta == 2:
        return float(np.sum((v - lam) ** 2))
    if beta == 0:
        q = v / lam
        return float(np.sum(q - np.log(q) - 1.0))
    return float(
        np.sum(v ** beta + (beta - 1.0) * lam ** beta - beta * v * lam ** (beta - 1.0))
        / (beta * (beta - 1.0))
    )


def sparse_nmf(v, p=None, *, rng=None, gpu_variant=False, mimic_matlab_flops=False):
    """[w, h, objective] = sparse_nmf(v, p)  --  src/sparse_nmf.m:1.

    `p` is a dict with the reference's field names.  Differences forced by the missing MATLAB
    runtime: random initial factors come from `rng` (a numpy Generator/RandomState-like object
    with `.random(shape)`), default ``np.random.RandomState(p.random_seed)``, because MATLAB's
    legacy ``rand('seed',s)`` stream (src/sparse_nmf.m:112-114) cannot be reproduced; parity
    runs always pass explicit ``init_w``/``init_h``.

    gpu_variant=True applies the deltas of src/sparse_nmf_GPU.m (no V floor :169 absent; cost
    always computed :261; objective vectors left zero :263-264; cost_check ignored).
    mimic_matlab_flops=True evaluates the `(...)*h'` products twice like the MATLAB expressions
    at :215-221 do (for CPU-baseline timing only; results are identical).
    """
    p = dict(p or {})
    v = np.asarray(v, dtype=np.float64)
    m, n = v.shape  # :71-72

    max_iter = int(p.get("max_iter", 100))  # :79-81
    random_seed = p.get("random_seed", 1)  # :83-85
    sparsity = p.get("sparsity", 0)  # :87-89
    conv_eps = float(p.get("conv_eps", 0))  # :91-93
    beta = _beta_from_cf(p)  # :95-110

    if rng is None:  # :112-114 (stand-in generator, see docstring)
        rng = np.random.RandomState(int(random_seed) if random_seed and random_seed > 0 else None)

    def _rand(a, b):
        return np.asarray(rng.random_sample((a, b)) if hasattr(rng, "random_sample") else rng.random((a, b)))

    # :116-131
    if "init_w" not in p or p["init_w"] is None:
        if "r" not in p:
            raise OracleError("Number of components or initialization must be given")
        r = int(p["r"])
        w = _rand(m, r)
    else:
        init_w = np.asarray(p["init_w"], dtype=np.float64)
        if init_w.shape[0] != m:
            raise OracleError("init_w rows must match v rows")
        ri = init_w.shape[1]
        if "r" in p and p["r"] is not None and ri < int(p["r"]):
            r = int(p["r"])
            w = np.concatenate([init_w, _rand(m, r - ri)], axis=1)
        else:
            r = ri
            w = init_w.copy()

    # :133-140
    init_h = p.get("init_h", None)
    if init_h is None:
        h = _rand(r, n)
    elif isinstance(init_h, str) and init_h == "ones":
        h = np.ones((r, n))
    else:
        h = np.array(init_h, dtype=np.float64)
        if h.shape != (r, n):
            raise OracleError("init_h must be r x n")

    # :142-148
    w_ind = np.asarray(p.get("w_update_ind", np.ones(r, bool))).astype(bool).reshape(-1)
    h_ind = np.asarray(p.get("h_update_ind", np.ones(r, bool))).astype(bool).reshape(-1)
    if w_ind.size != r or h_ind.size != r:
        raise OracleError("update index vectors must have r entries")

    # :150-155  sparsity per matrix entry
    sp = np.asarray(sparsity, dtype=np.float64)
    if sp.size == 1:
        S = np.full((r, n), float(sp.reshape(-1)[0]))
    elif sp.ndim == 1 or (sp.ndim == 2 and sp.shape[1] == 1):
        S = np.repeat(sp.reshape(-1, 1), n, axis=1)
    else:
        S = sp.copy()

    # :157-160  normalise columns of W, rescale H (applies to a user init_h too)
    wn = np.sqrt(np.sum(w ** 2, axis=0))
    w = w / wn
    h = h * wn[:, None]

    flr = FLR  # :166
    # max(.,flr) is np.fmax throughout: MATLAB's max skips NaN (max(NaN, flr) = flr)
    lam = np.fmax(w @ h, flr)  # :167
    last_cost = np.inf  # :168
    if not gpu_variant:
        v = np.fmax(v, flr)  # :169 (CPU file only)

    div_hist = np.zeros(max_iter)  # :171-173
    cost_hist = np.zeros(max_iter)

    update_h = int(h_ind.sum())  # :178
    update_w = int(w_ind.sum())  # :179
    if update_h > 0:
        # :192,:197,:202 -- `S` is r x n while the contraction has sum(h_ind) rows, so a partial
        # h_update_ind is a MATLAB dimension error unless the user-supplied S has that many rows.
        if S.shape[0] != update_h or S.shape[1] != n:
            raise OracleError("sparsity rows must match the updated rows of h")
        if S.shape[0] != r:
            # cost at :261 multiplies S (update_h rows) with h (r rows) -> size error when checked
            if gpu_variant or p.get("cost_check", None):
                raise OracleError("sparsity .* h size mismatch")
    elif S.shape != (r, n):
        if gpu_variant or p.get("cost_check", None):
            raise OracleError("sparsity .* h size mismatch")

    if not gpu_variant and "cost_check" not in p:
        # :260 reads p.cost_check without a default -> MATLAB "Reference to non-existent field"
        raise OracleError("Reference to non-existent field 'cost_check'.")
    cost_check = True if gpu_variant else bool(p["cost_check"])

    # p.display (:162-164 default 0): the console output, written where the reference writes it
    import sys
    display = p.get("display", 0) != 0
    put = sys.stdout.write
    pstr = ""  # :70 str = []
    if display and not gpu_variant:
        put("Performing sparse NMF with beta-divergence, beta=%.1f\n" % beta)  # :181-183

    n_iter = max_iter
    for it in range(1, max_iter + 1):  # :186
        # ---- H updates :189-208
        if update_h > 0:
            wh = w[:, h_ind]
            if beta == 1:
                dph = np.sum(wh, axis=0)[:, None] + S  # :192
                dph = np.fmax(dph, flr)  # :193
                dmh = wh.T @ (v / lam)  # :194
                h[h_ind, :] = h[h_ind, :] * dmh / dph  # :195
            elif beta == 2:
                dph = wh.T @ lam + S  # :197
                dph = np.fmax(dph, flr)
                dmh = wh.T @ v  # :199
                h[h_ind, :] = h[h_ind, :] * dmh / dph
            else:
                dph = wh.T @ lam ** (beta - 1.0) + S  # :202
                dph = np.fmax(dph, flr)
                dmh = wh.T @ (v * lam ** (beta - 2.0))  # :204
                h[h_ind, :] = h[h_ind, :] * dmh / dph
            lam = np.fmax(w @ h, flr)  # :207

        # ---- W updates :212-244
        if update_w > 0:
            ww = w[:, w_ind]
            hw = h[w_ind, :]
            if beta == 1:
                G = (v / lam) @ hw.T  # :217,:219
                if mimic_matlab_flops:
                    G = (v / lam) @ hw.T
                s = np.sum(hw, axis=1)[None, :]  # :215
                dpw = s + np.sum(G * ww, axis=0)[None, :] * ww  # :215-217
                dpw = np.fmax(dpw, flr)  # :218
                dmw = G + np.sum(s * ww, axis=0)[None, :] * ww  # :219-221
            elif beta == 2:
                P = lam @ hw.T  # :224,:228
                Q = v @ hw.T  # :225,:227
                if mimic_matlab_flops:
                    P = lam @ hw.T
                    Q = v @ hw.T
                dpw = P + np.sum(Q * ww, axis=0)[None, :] * ww
                dpw = np.fmax(dpw, flr)
                dmw = Q + np.sum(P * ww, axis=0)[None, :] * ww
            else:
                P = lam ** (beta - 1.0) @ hw.T  # :231,:238
                Q = (v * lam ** (beta - 2.0)) @ hw.T  # :233,:236
                if mimic_matlab_flops:
                    P = lam ** (beta - 1.0) @ hw.T
                    Q = (v * lam ** (beta - 2.0)) @ hw.T
                dpw = P + np.sum(Q * ww, axis=0)[None, :] * ww
                dpw = np.fmax(dpw, flr)
                dmw = Q + np.sum(P * ww, axis=0)[None, :] * ww
            w[:, w_ind] = ww * dmw / dpw  # :222,:229,:239
            w = w / np.sqrt(np.sum(w ** 2, axis=0))  # :242 -- ALL columns, H not rescaled
            lam = np.fmax(w @ h, flr)  # :243

        # ---- objective :248-258 (always evaluated)
        div = divergence(v, lam, beta)

        if cost_check:  # :260
            cost = div + float(np.sum(S * h))  # :261
            if not gpu_variant:
                div_hist[it - 1] = div  # :263-264
                cost_hist[it - 1] = cost
            if display:
                if gpu_variant:
                    put("iteration %d div = %.3e cost = %.3e\n" % (it, div, cost))  # sparse_nmf_GPU.m:266-268
                else:
                    put("\b" * len(pstr))  # :267
                    pstr = "iteration %d div = %.3e cost = %.3e" % (it, div, cost)  # :268
                    put(pstr)  # :269
            if it > 1 and conv_eps > 0:  # :273
                with np.errstate(divide="ignore", invalid="ignore"):  # MATLAB: x/0 = Inf, 0/0 = NaN (NaN < eps is false)
                    e = np.float64(abs(cost - last_cost)) / np.float64(last_cost)  # :274
                if e < conv_eps:  # :275
                    if display:  # :276-278 (sparse_nmf_GPU.m:274 prints it unconditionally; not restated)
                        put("Convergence reached, aborting iteration\n")
                    if not gpu_variant:
                        div_hist = div_hist[:it]  # :279-280
                        cost_hist = cost_hist[:it]
                    n_iter = it
                    break
            last_cost = cost  # :284

    if display and not gpu_variant:
        put("\\nMax Iteration reached, aborting iteration\\n\n")  # :288-290: disp of a single-quoted string, after a stop too
    objective = {"div": div_hist, "cost": cost_hist, "n_iter": n_iter}
    return w, h, objective


def run_basis_dnmf_solves(Y, X, D, B, R_x, R_d, p):
    """The three solver calls of run_basis_DNMF.m:36-55 on already-formed spectrograms.

    Y, X, D: F x T power/magnitude features (run_basis_DNMF.m:13-34 produce them; the STFT front
    end is outside this path).  Returns (B_hat, A_hat).
    """
    p = dict(p)
    # :37-40  H-only on the mixture
    p["w_update_ind"] = np.zeros(R_x + R_d, bool)
    p["h_update_ind"] = np.ones(R_x + R_d, bool)
    p["init_w"] = B
    p.pop("init_h", None)
    _, A_hat, _ = sparse_nmf(Y, p)
    # :43-47  W-only on clean speech
    p["w_update_ind"] = np.ones(R_x, bool)
    p["h_update_ind"] = np.zeros(R_x, bool)
    p["init_w"] = B[:, :R_x]
    p["init_h"] = A_hat[:R_x, :]
    B_hat_x, _, _ = sparse_nmf(X, p)
    # :49-53  W-only on noise
    p["w_update_ind"] = np.ones(R_d, bool)
    p["h_update_ind"] = np.zeros(R_d, bool)
    p["init_w"] = B[:, R_x:R_x + R_d]
    p["init_h"] = A_hat[R_x:R_x + R_d, :]
    B_hat_d, _, _ = sparse_nmf(D, p)
    return np.concatenate([B_hat_x, B_hat_d], axis=1), A_hat  # :55


def synth_problem(F, T, r, *, seed_data=0, seed_init=1, scale="unit", r_true=None):
    """Deterministic synthetic |STFT|-like input.  Returns V, W0, H0 (fp64)."""
    rd = np.random.default_rng(seed_data)
    r_true = r if r_true is None else r_true
    Wt = rd.gamma(0.5, 1.0, size=(F, r_true))
    Ht = rd.gamma(0.3, 1.0, size=(r_true, T))
    V = Wt @ Ht
    if scale == "power":
        V = V * (1e10 / V.max())
    V = V + 1e-9
    ri = np.random.default_rng(seed_init)
    W0 = ri.random((F, r))
    H0 = ri.random((r, T))
    return V, W0, H0
