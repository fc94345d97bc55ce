"""Rank reduction of an over-complete exemplar dictionary by clustering, run_basis_train.m:118-129:

    [~, ~, ~, D] = kmeans(B_Mel_init', R, 'distance', 'cityblock', 'emptyaction', 'singleton', ...
                          'onlinephase', 'off', 'start', 'cluster');
    [~, Dmin_idx] = min(D);        % per cluster: the observation nearest to its centroid

This is host-side control logic of the training DRIVER (MATLAB runs it on the CPU too; the data are cluster_buff*R
basis vectors of a few dozen to a few hundred Mel bins), not part of the solver hot path, so it is plain NumPy.  It
restates the documented behaviour of MATLAB's kmeans for exactly these options:
  * 'cityblock': L1 distance, a centroid is the component-wise MEDIAN of its members;
  * 'start','cluster': a preliminary clustering on a random 10 % subsample (itself started from k random observations)
    when that subsample has more than k observations, else k observations of X at random;
  * 'onlinephase','off': batch updates only (assign all, then move all), at most 100 iterations (MaxIter default);
  * 'emptyaction','singleton': an empty cluster is re-created from the one observation furthest from its centroid
    (over all observations; should that one be its cluster's only member: the first member of the first cluster with two).
MATLAB's random stream cannot be reproduced, so the draws come from a seeded NumPy generator: the
result is the reference's ALGORITHM on its data, not MATLAB's bits.  tests/test_kmeans.py checks it against an
independent loop restatement (oracle/kmeans_oracle.py) and by its invariants.
"""
from __future__ import annotations

import numpy as np


def _cityblock(X, C):
    """n x k matrix of L1 distances (chunked over the observations: n*k*p floats would be large for big dictionaries)."""
    n, k = X.shape[0], C.shape[0]
    D = np.empty((n, k))
    step = max(1, int(4e6 // max(1, k * X.shape[1])))
    for i0 in range(0, n, step):
        D[i0:i0 + step] = np.abs(X[i0:i0 + step, None, :] - C[None, :, :]).sum(-1)
    return D


def _batch_phase(X, C, max_iter):
    """Batch k-medians from the centroids C: returns (idx, C, D, n_iter, totals per iteration)."""
    n, k = X.shape[0], C.shape[0]
    C = np.array(C, dtype=np.float64, copy=True)
    idx = np.full(n, -1)
    totals = []
    it = 0
    for it in range(1, max_iter + 1):
        D = _cityblock(X, C)
        new = D.argmin(1)
        own = D[np.arange(n), new]
        # 'singleton' as MATLAB's kmeans implements it in the batch phase: every empty cluster is re-created from the observation
        # furthest from its current centroid, [dlarge, lonely] = max(d) over ALL observations (first maximum); "in the very
        # unusual event that the cluster had only one member, pick any other non-singleton point": from = find(m > 1, 1,
        # 'first'), lonely = find(idx == from, 1, 'first') -- so the repair never empties its donor (n >= k: while a cluster
        # is empty another holds two).
        counts = np.bincount(new, minlength=k)
        for j in np.flatnonzero(counts == 0):
            far = int(own.argmax())
            frm = int(new[far])
            if counts[frm] < 2:
                frm = int(np.flatnonzero(counts > 1)[0])
                far = int(np.flatnonzero(new == frm)[0])
            counts[frm] -= 1
            new[far] = j
            counts[j] = 1
            own[far] = 0.0
        totals.append(float(own.sum()))
        if np.array_equal(new, idx):
            break
        idx = new
        for j in range(k):
            C[j] = np.median(X[idx == j], axis=0)
    D = _cityblock(X, C)
    return idx, C, D, it, totals


def kmeans_cityblock(X, k, *, seed=1, max_iter=100):
    """idx, C, sumd, D as MATLAB's [idx, C, sumd, D] = kmeans(X, k, ...) with the options of run_basis_train.m:120-123.
    X: n x p observations in rows.  idx is 0-based."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    k = int(k)
    if not 1 <= k <= n:
        raise ValueError("kmeans: need 1 <= k <= number of observations")
    rs = np.random.RandomState(seed)
    n_sub = int(np.floor(0.1 * n))
    if n_sub > k:  # 'start','cluster'
        sub = X[rs.choice(n, n_sub, replace=False)]
        C0 = _batch_phase(sub, sub[rs.choice(n_sub, k, replace=False)], max_iter)[1]
    else:
        C0 = X[rs.choice(n, k, replace=False)]
    idx, C, D, _, _ = _batch_phase(X, C0, max_iter)
    sumd = np.array([D[idx == j, j].sum() for j in range(k)])
    return idx, C, sumd, D


def reduce_rank(B_Mel, B_DFT, A_DFT, A_Mel, R, *, seed=1):
    """run_basis_train.m:118-129: keep, for each of R clusters of the Mel basis vectors, the vector nearest to the
    cluster's centroid -- the same columns of both dictionaries and the same rows of both activation matrices."""
    _, C, _, D = kmeans_cityblock(np.asarray(B_Mel).T, R, seed=seed)  # :120-123 (observations = basis vectors)
    if not (np.isfinite(C).all() and np.isfinite(D).all()):
        raise ValueError("kmeans: non-finite centroid (empty cluster)")
    keep = D.argmin(0)                                                # :124  [~, Dmin_idx] = min(D)
    return B_Mel[:, keep], B_DFT[:, keep], A_DFT[keep, :], A_Mel[keep, :], keep
