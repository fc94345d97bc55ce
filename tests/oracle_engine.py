"""Test-only engine with the step interface of se_snmf_nat_amd.api.Plan, backed by fp64 NumPy.

Used to drive se_snmf_nat_amd.dist.ShardedLoop over gloo on CPU: it pins the sharding algebra
(what is summed across ranks, when the objective of which iterate is formed, how the stop index
is recovered) against the unsharded oracle.  Same statistics layout idea as the HIP plan:
[M0 (F*r) | M1 (F*r, beta != 1) | s (r) | div | sh].  Follows src/sparse_nmf.m:186-286.
"""
import numpy as np

from oracle.sparse_nmf_oracle import FLR, divergence


class OracleEngine:
    def __init__(self, v, w0, h0, *, beta, sparsity, max_iter, conv_eps, cost_check, w_ind=None, h_ind=None):
        self.v = np.maximum(np.asarray(v, np.float64), FLR)
        w = np.array(w0, np.float64)
        h = np.array(h0, np.float64)
        wn = np.sqrt((w ** 2).sum(0))
        self.w = w / wn
        self.h = h * wn[:, None]
        self.h_prev = self.h
        self.F, self.r = w.shape
        self.beta, self.lam_s, self.max_iter = float(beta), float(sparsity), max_iter
        self.conv_eps, self.cost_check = conv_eps, cost_check
        self.w_ind = np.ones(self.r, bool) if w_ind is None else np.asarray(w_ind, bool)
        self.upd_h = True if h_ind is None else bool(np.asarray(h_ind).any())
        self.upd_w = bool(self.w_ind.any())
        self.n_mat = 1 if beta == 1 else 2
        self.it = 0
        self.stop = False
        self.n_iter = 0
        self.div_hist, self.cost_hist = [], []
        self._obj_local = (0.0, 0.0)

    def stats_len(self):
        return self.n_mat * self.F * self.r + self.r + 2

    def _lam(self):
        return np.maximum(self.w @ self.h, FLR)

    def hstep(self):
        if self.stop:
            return
        lam = self._lam()
        # objective of the CURRENT iterate (it) -- formed before H moves
        self._obj_local = (divergence(self.v, lam, self.beta), self.lam_s * self.h.sum())
        if not self.upd_h:
            return
        b, v, w = self.beta, self.v, self.w
        if b == 1:
            dph = np.maximum(w.sum(0)[:, None] + self.lam_s, FLR)
            dmh = w.T @ (v / lam)
        elif b == 2:
            dph = np.maximum(w.T @ lam + self.lam_s, FLR)
            dmh = w.T @ v
        else:
            dph = np.maximum(w.T @ lam ** (b - 1) + self.lam_s, FLR)
            dmh = w.T @ (v * lam ** (b - 2))
        self.h_prev = self.h
        self.h = self.h * dmh / dph

    def _view(self, ptr_or_arr):
        return ptr_or_arr

    def wstats(self, stats):
        if self.stop:
            return
        st = stats
        st[:] = 0
        n = self.F * self.r
        if self.upd_w:
            lam = self._lam()
            b, v, h = self.beta, self.v, self.h
            if b == 1:
                st[:n] = ((v / lam) @ h.T).reshape(-1, order="F")
                st[n * self.n_mat:n * self.n_mat + self.r] = h.sum(1)
            elif b == 2:
                st[:n] = (v @ h.T).reshape(-1, order="F")
                st[n:2 * n] = (lam @ h.T).reshape(-1, order="F")
            else:
                st[:n] = ((v * lam ** (b - 2)) @ h.T).reshape(-1, order="F")
                st[n:2 * n] = (lam ** (b - 1) @ h.T).reshape(-1, order="F")
        st[-2], st[-1] = self._obj_local

    def _check(self, st, it):
        div, cost = st[-2], st[-2] + st[-1]
        stopnow = False
        if it > 1 and self.conv_eps > 0:
            last = self.cost_hist[it - 2]
            stopnow = abs(cost - last) / last < self.conv_eps
        self.div_hist.append(div)
        self.cost_hist.append(cost)
        self.n_iter = it
        if stopnow:
            self.stop = True
        return stopnow

    def wapply(self, stats):
        if self.stop:
            self.it += 1
            return
        st = stats
        j = self.it + 1
        if self.cost_check and j > 1:
            if self._check(st, j - 1):
                # result is iterate j-1: H before this iteration's hstep
                self.h = self.h_prev
                self.it = j
                return
        if self.upd_w:
            n = self.F * self.r
            Q = st[:n].reshape(self.F, self.r, order="F")
            if self.n_mat == 2:
                P = st[n:2 * n].reshape(self.F, self.r, order="F")
            else:
                P = np.broadcast_to(st[n:n + self.r][None, :], (self.F, self.r))
            w = self.w
            ww, Qw, Pw = w[:, self.w_ind], Q[:, self.w_ind], P[:, self.w_ind]
            dpw = np.maximum(Pw + (Qw * ww).sum(0)[None, :] * ww, FLR)
            dmw = Qw + (Pw * ww).sum(0)[None, :] * ww
            w = w.copy()
            w[:, self.w_ind] = ww * dmw / dpw
            self.w = w / np.sqrt((w ** 2).sum(0))
        self.it = j

    def objstats(self, stats):
        if self.stop:
            return
        stats[:] = 0
        lam = self._lam()
        stats[-2] = divergence(self.v, lam, self.beta)
        stats[-1] = self.lam_s * self.h.sum()

    def objapply(self, stats):
        if self.stop or self.it < 1:
            return
        self._check(stats, self.it)

    def stopped(self):
        return self.stop
