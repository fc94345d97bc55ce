"""TEST INFRASTRUCTURE ONLY (never imported by the product).  Loop restatement of the clustering call of
run_basis_train.m:120-124 -- MATLAB kmeans with 'distance','cityblock', 'emptyaction','singleton', 'onlinephase','off',
'start','cluster' -- written independently of se_snmf_nat_amd/kmeans.py (pure Python loops, no broadcasting) so that the
two can be compared draw for draw: both take their random choices from numpy.random.RandomState(seed) in the same order
(subsample, start observations).  Parity with MATLAB itself is unpinned: its random stream cannot be reproduced."""
import numpy as np


def _dist(x, c):
    s = 0.0
    for a, b in zip(x, c):
        s += abs(a - b)
    return s


def _batch(X, C, max_iter):
    n, k = len(X), len(C)
    C = [list(c) for c in C]
    idx = [-1] * n
    for _ in range(max_iter):
        D = [[_dist(X[i], C[j]) for j in range(k)] for i in range(n)]
        new = [min(range(k), key=lambda j: (D[i][j], j)) for i in range(n)]
        own = [D[i][new[i]] for i in range(n)]
        counts = [new.count(j) for j in range(k)]
        for j in range(k):
            if counts[j] == 0:
                # 'emptyaction','singleton' as MATLAB's kmeans documents and implements it (batch phase): the observation
                # furthest from its current centroid founds the new cluster -- [dlarge, lonely] = max(d) over ALL observations --
                # and "in the very unusual event that the cluster had only one member, pick any other non-singleton point":
                # from = find(m > 1, 1, 'first'); lonely = find(idx == from, 1, 'first').
                far = 0
                for i in range(1, n):
                    if own[i] > own[far]:
                        far = i
                frm = new[far]
                if counts[frm] < 2:
                    frm = next(c for c in range(k) if counts[c] > 1)
                    far = next(i for i in range(n) if new[i] == frm)
                counts[frm] -= 1
                new[far] = j
                counts[j] = 1
                own[far] = 0.0
        if new == idx:
            break
        idx = new
        for j in range(k):
            members = [X[i] for i in range(n) if idx[i] == j]
            C[j] = [float(np.median([m[d] for m in members])) for d in range(len(X[0]))]
    D = [[_dist(X[i], C[j]) for j in range(k)] for i in range(n)]
    return idx, C, D


def kmeans_cityblock(X, k, seed=1, max_iter=100):
    X = [list(map(float, row)) for row in np.asarray(X, dtype=np.float64)]
    n = len(X)
    rs = np.random.RandomState(seed)
    n_sub = int(np.floor(0.1 * n))
    if n_sub > k:
        sub = [X[i] for i in rs.choice(n, n_sub, replace=False)]
        C0 = _batch(sub, [sub[i] for i in rs.choice(n_sub, k, replace=False)], max_iter)[1]
    else:
        C0 = [X[i] for i in rs.choice(n, k, replace=False)]
    idx, C, D = _batch(X, C0, max_iter)
    return np.array(idx), np.array(C), np.array(D)
