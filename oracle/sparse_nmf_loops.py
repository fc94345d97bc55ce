"""Second, independently written restatement of src/sparse_nmf.m -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/sparse_nmf_oracle.py).  This file exists to catch transcription
errors in the vectorised oracle: it is written element by element, straight from the MATLAB
expressions, with pure-Python loops and no matrix products, so it is only usable on tiny cases.
It supports exactly what the cross-check needs: explicit init_w/init_h, boolean masks, scalar
sparsity, any beta.  Reference lines cited per block.
"""
from __future__ import annotations

import math

FLR = 1e-9  # src/sparse_nmf.m:166


def _lam(w, h, m, n, r):
    # src/sparse_nmf.m:167,:207,:243   lambda = max(w*h, flr)
    out = [[0.0] * n for _ in range(m)]
    for f in range(m):
        for t in range(n):
            acc = 0.0
            for k in range(r):
                acc += w[f][k] * h[k][t]
            out[f][t] = acc if acc > FLR else FLR
    return out


def sparse_nmf_loops(v, init_w, init_h, beta, sparsity, max_iter, conv_eps, w_ind, h_ind, cost_check=True):
    m, n = len(v), len(v[0])
    r = len(init_w[0])
    w = [row[:] for row in init_w]
    h = [row[:] for row in init_h]
    # :157-160
    for k in range(r):
        nk = math.sqrt(sum(w[f][k] ** 2 for f in range(m)))
        for f in range(m):
            w[f][k] /= nk
        for t in range(n):
            h[k][t] *= nk
    v = [[x if x > FLR else FLR for x in row] for row in v]  # :169
    lam = _lam(w, h, m, n, r)
    last = math.inf
    divs, costs = [], []
    upd_h = any(h_ind)
    upd_w = any(w_ind)
    if upd_h and not all(h_ind):
        raise ValueError("partial h_update_ind is a size error in the reference (:192)")
    for it in range(1, max_iter + 1):
        if upd_h:  # :189-208
            for k in range(r):
                for t in range(n):
                    if beta == 1:
                        den = sum(w[f][k] for f in range(m)) + sparsity  # :192
                        num = sum(w[f][k] * (v[f][t] / lam[f][t]) for f in range(m))  # :194
                    elif beta == 2:
                        den = sum(w[f][k] * lam[f][t] for f in range(m)) + sparsity  # :197
                        num = sum(w[f][k] * v[f][t] for f in range(m))  # :199
                    else:
                        den = sum(w[f][k] * lam[f][t] ** (beta - 1) for f in range(m)) + sparsity  # :202
                        num = sum(w[f][k] * v[f][t] * lam[f][t] ** (beta - 2) for f in range(m))  # :204
                    den = den if den > FLR else FLR
                    # all (k,t) use the OLD lambda, so write into a scratch copy
                    h[k][t] = (h[k][t] * num / den, )  # tuple marks "new"
            for k in range(r):
                for t in range(n):
                    h[k][t] = h[k][t][0]
            lam = _lam(w, h, m, n, r)
        if upd_w:  # :212-244
            ks = [k for k in range(r) if w_ind[k]]
            A = {}
            Bm = {}
            for k in ks:
                for f in range(m):
                    if beta == 1:
                        A[f, k] = sum(v[f][t] / lam[f][t] * h[k][t] for t in range(n))  # G :217
                    elif beta == 2:
                        A[f, k] = sum(v[f][t] * h[k][t] for t in range(n))  # Q :225
                        Bm[f, k] = sum(lam[f][t] * h[k][t] for t in range(n))  # P :224
                    else:
                        A[f, k] = sum(v[f][t] * lam[f][t] ** (beta - 2) * h[k][t] for t in range(n))
                        Bm[f, k] = sum(lam[f][t] ** (beta - 1) * h[k][t] for t in range(n))
            neww = {}
            for k in ks:
                if beta == 1:
                    s = sum(h[k][t] for t in range(n))  # :215
                    cA = sum(A[f, k] * w[f][k] for f in range(m))
                    cS = sum(s * w[f][k] for f in range(m))
                    for f in range(m):
                        dpw = s + cA * w[f][k]
                        dpw = dpw if dpw > FLR else FLR
                        dmw = A[f, k] + cS * w[f][k]
                        neww[f, k] = w[f][k] * dmw / dpw
                else:
                    cQ = sum(A[f, k] * w[f][k] for f in range(m))
                    cP = sum(Bm[f, k] * w[f][k] for f in range(m))
                    for f in range(m):
                        dpw = Bm[f, k] + cQ * w[f][k]
                        dpw = dpw if dpw > FLR else FLR
                        dmw = A[f, k] + cP * w[f][k]
                        neww[f, k] = w[f][k] * dmw / dpw
            for (f, k), x in neww.items():
                w[f][k] = x
            for k in range(r):  # :242 all columns
                nk = math.sqrt(sum(w[f][k] ** 2 for f in range(m)))
                for f in range(m):
                    w[f][k] /= nk
            lam = _lam(w, h, m, n, r)
        # :248-258
        div = 0.0
        for f in range(m):
            for t in range(n):
                x, l = v[f][t], lam[f][t]
                if beta == 1:
                    div += x * math.log(x / l) - x + l
                elif beta == 2:
                    div += (x - l) ** 2
                elif beta == 0:
                    div += x / l - math.log(x / l) - 1
                else:
                    div += (x ** beta + (beta - 1) * l ** beta - beta * x * l ** (beta - 1)) / (beta * (beta - 1))
        if cost_check:
            cost = div + sparsity * sum(h[k][t] for k in range(r) for t in range(n))  # :261
            divs.append(div)
            costs.append(cost)
            if it > 1 and conv_eps > 0 and abs(cost - last) / last < conv_eps:  # :273-275
                break
            last = cost
    return w, h, divs, costs
