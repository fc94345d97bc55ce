"""CPU oracle for the missing-data-imputation variants of the solver.
TEST INFRASTRUCTURE ONLY -- the product never imports this module.

PARITY UNPINNED: the reference is MATLAB and cannot run here (see oracle/sparse_nmf_oracle.py).
fp64 NumPy restatement of
  src/snmf_mdi.m:71-312      [v_MDI, h, objective] = snmf_mdi(v, Dm, p)      binary observed/missing mask
  src/snmf_mdi_Sm.m:71-318   [v_MDI, h, objective] = snmf_mdi_Sm(v, Sm, p)   soft mask in [0,1]
The two files are the solver of src/sparse_nmf.m with (a) the masked start v = max(v.*M, flr) (:175),
(b) a re-imputation of v after every iteration, v = max(v.*M + max(w*h,flr).*(1-M), flr) (:251-254 /
:251-260), BEFORE the objective is evaluated, and (c) a gain-matched final imputation (:296-306 /
:302-309).  Parameter quirks restated as written: the sparsity weight and the stopping threshold are read
from p.sparsity_mdi / p.conv_eps_mdi, and their defaults are only installed when p.sparsity / p.conv_eps
are ABSENT (:87-93) -- a struct that has `sparsity` but no `sparsity_mdi` is a missing-field error.
MATLAB's max skips NaN (np.fmax).  Random initial factors: explicit init_w / init_h, or a stand-in
RandomState (MATLAB's legacy generator is unrecoverable).
"""
from __future__ import annotations

import numpy as np

FLR = 1e-9


class MdiOracleError(Exception):
    pass


def _divergence(v, lam, beta):
    """:263-274"""
    if beta == 1:
        return float(np.sum(v * np.log(v / lam) - v + lam))
    if beta == 2:
        return float(np.sum((v - lam) ** 2))
    if beta == 0:
        return float(np.sum(v / lam - np.log(v / lam) - 1))
    return float(np.sum(v ** beta + (beta - 1) * lam ** beta - beta * v * lam ** (beta - 1)) / (beta * (beta - 1)))


def snmf_mdi(v, mask, p, rng=None):
    """src/snmf_mdi.m / src/snmf_mdi_Sm.m (the mask decides: `.*~Dm` == `.*(1-Sm)` for a 0/1 mask).
    Returns (v_MDI, h, objective) with objective = dict(div, cost, n_iter)."""
    p = dict(p or {})
    v = np.asarray(v, dtype=np.float64)
    M = np.asarray(mask, dtype=np.float64)
    m, n = v.shape
    if M.shape != v.shape:
        raise MdiOracleError("mask must have the size of v")
    max_iter = int(p.get("max_iter", 100))  # :83-85
    random_seed = p.get("random_seed", 1)
    if "sparsity" not in p:  # :87-89
        p.setdefault("sparsity_mdi", 0)
    if "conv_eps" not in p:  # :91-93
        p.setdefault("conv_eps_mdi", 0)
    for fld in ("sparsity_mdi", "conv_eps_mdi", "cost_check"):
        if fld not in p:
            raise KeyError(f"Reference to non-existent field '{fld}'.")
    cf = p.get("cf", "kl")  # :95-110
    beta = {"is": 0.0, "kl": 1.0, "ed": 2.0}.get(cf, float(p.get("beta", 1.0)))
    if rng is None:
        rng = np.random.RandomState(int(random_seed) if random_seed and random_seed > 0 else None)
    if p.get("init_w") is None:  # :116-131
        if "r" not in p:
            raise MdiOracleError("Number of components or initialization must be given")
        r = int(p["r"])
        w = rng.random_sample((m, r))
    else:
        w = np.array(p["init_w"], dtype=np.float64)
        r = w.shape[1]
        if p.get("r") is not None and r < int(p["r"]):
            w = np.concatenate([w, rng.random_sample((m, int(p["r"]) - r))], axis=1)
            r = int(p["r"])
    init_h = p.get("init_h")  # :133-140
    if init_h is None:
        h = rng.random_sample((r, n))
    elif isinstance(init_h, str) and init_h == "ones":
        h = np.ones((r, n))
    else:
        h = np.array(init_h, dtype=np.float64)
    w_ind = np.asarray(p.get("w_update_ind", np.ones(r, bool))).astype(bool).reshape(-1)  # :142-148
    h_ind = np.asarray(p.get("h_update_ind", np.ones(r, bool))).astype(bool).reshape(-1)
    sp = np.asarray(p["sparsity_mdi"], dtype=np.float64)  # :150-160
    if sp.size == 1:
        S = np.full((r, n), float(sp.reshape(-1)[0]))
    elif sp.ndim == 1 or sp.shape[1] == 1:
        S = np.repeat(sp.reshape(-1, 1), n, axis=1)
    else:
        S = sp.copy()
    conv_eps = float(p["conv_eps_mdi"])

    wn = np.sqrt(np.sum(w ** 2, axis=0))  # :163-165
    w = w / wn
    h = h * wn[:, None]
    flr = FLR
    lam = np.fmax(w @ h, flr)  # :172
    last_cost = np.inf
    v = np.fmax(v * M, flr)  # :175 masked start
    div_hist = np.zeros(max_iter)
    cost_hist = np.zeros(max_iter)
    update_h, update_w = int(h_ind.sum()), int(w_ind.sum())
    n_iter = max_iter
    for it in range(1, max_iter + 1):
        if update_h > 0:  # :190-209
            wh = w[:, h_ind]
            if beta == 1:
                dph = np.fmax(np.sum(wh, axis=0)[:, None] + S, flr)
                dmh = wh.T @ (v / lam)
                h[h_ind, :] = h[h_ind, :] * dmh / dph
            elif beta == 2:
                dph = np.fmax(wh.T @ lam + S, flr)
                dmh = wh.T @ v
                h[h_ind, :] = h[h_ind, :] * dmh / dph
            else:
                dph = np.fmax(wh.T @ lam ** (beta - 1) + S, flr)
                dmh = wh.T @ (v * lam ** (beta - 2))
                h[h_ind, :] = h[h_ind, :] * dmh / dph
            lam = np.fmax(w @ h, flr)
        if update_w > 0:  # :213-248
            hw, ww = h[w_ind, :], w[:, w_ind]
            if beta == 1:
                G = (v / lam) @ hw.T
                s = np.sum(hw, axis=1)[None, :]
                dpw = np.fmax(s + np.sum(G * ww, axis=0)[None, :] * ww, flr)
                dmw = G + np.sum(s * ww, axis=0)[None, :] * ww
            else:
                if beta == 2:
                    P, Q = lam @ hw.T, v @ hw.T
                else:
                    P, Q = lam ** (beta - 1) @ hw.T, (v * lam ** (beta - 2)) @ hw.T
                dpw = np.fmax(P + np.sum(Q * ww, axis=0)[None, :] * ww, flr)
                dmw = Q + np.sum(P * ww, axis=0)[None, :] * ww
            w[:, w_ind] = ww * dmw / dpw
            w = w / np.sqrt(np.sum(w ** 2, axis=0))
            lam = np.fmax(w @ h, flr)
        v_est = np.fmax(w @ h, flr)  # :251-254: estimated missing part
        v = np.fmax(v * M + v_est * (1 - M), flr)
        div = _divergence(v, lam, beta)  # :257-268
        if p["cost_check"]:  # :270-295
            cost = div + float(np.sum(S * h))
            div_hist[it - 1] = div
            cost_hist[it - 1] = cost
            if it > 1 and conv_eps > 0:
                with np.errstate(divide="ignore", invalid="ignore"):
                    e = np.float64(abs(cost - last_cost)) / np.float64(last_cost)
                if e < conv_eps:
                    n_iter = it
                    div_hist, cost_hist = div_hist[:it], cost_hist[:it]
                    break
            last_cost = cost
    v_est = np.fmax(w @ h, flr)  # :298-306 gain-matched final imputation
    Nt = np.sum(v * M, axis=0) / np.fmax(np.sum(v_est * M, axis=0), flr)
    v_mdi = np.fmax(v * M + Nt[None, :] * v_est * (1 - M), flr)
    return v_mdi, h, {"div": div_hist, "cost": cost_hist, "n_iter": n_iter, "w": w}
