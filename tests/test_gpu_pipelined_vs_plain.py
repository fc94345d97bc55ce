"""The pipelined kernels of the headline path against the plain ones, bit for bit, through the product path.

k_hstep_rp (role pipeline: loader waves with buffer loads / stores and scalar offsets, progress-slot signalling) and
k_wstats with LDS-DMA loader waves compute exactly what the barrier-phased k_hstep and the synchronously staging
k_wstats compute (same MFMA order per tile), so H after H-only iterations must agree in every bit and W after full
iterations to the summation order of the row sums.  This is the check that found the two synchronisation bugs and the
buffer-store hazard of round 2 (profiles/r02_experiments.md): the shapes are the ones that showed them -- few row
tiles with several tiles per workgroup, short H rows (rp < 256: lanes past the row duplicate lane 0), a V block whose
last cell straddles the end.  SNMF_HSTEP_RP / SNMF_WSTATS_NL are read when a plan is created, so both variants run in
this one process.  (src/sparse_nmf.m:189-208 and :215-239 are the updates both variants implement.)
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(65, 70, 12000), (129, 70, 12000), (65, 32, 12000), (64, 70, 12000), (97, 96, 20000), (257, 40, 20000),
          (33, 8, 30000), (161, 200, 9000), (257, 256, 30000), (225, 100, 17000), (513, 64, 12000), (385, 100, 12000)]  # (the last two: four / four row groups in k_wstats)


def _run(ctx, V, W0, H0, r, *, h_only, iters):
    from se_snmf_nat_amd import Plan
    F, T = V.shape
    kw = dict(w_update_ind=np.zeros(r, bool)) if h_only else {}
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=iters, conv_eps=0.0, cost_check=True, sparsity=1.0, **kw)
    pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
    out = (pl.get_h(np.float32), pl.get_w(), pl.describe())
    pl.close()
    return out


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "F%d_r%d_T%d" % s)
def test_pipelined_kernels_equal_plain_kernels(gpu_ctx, shape, monkeypatch):
    F, r, T = shape
    rs = np.random.default_rng(F * 1000 + r)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    monkeypatch.delenv("SNMF_HSTEP_RP", raising=False)
    monkeypatch.delenv("SNMF_WSTATS_NL", raising=False)
    h_new, _, geo = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    _, w_new, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    assert "k_hstep_rp" in geo  # the pipelined path is what ran
    monkeypatch.setenv("SNMF_HSTEP_RP", "0")
    monkeypatch.setenv("SNMF_WSTATS_NL", "0")
    h_old, _, geo_old = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    _, w_old, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    assert "k_hstep_rp" not in geo_old
    assert np.array_equal(h_new, h_old)
    assert np.abs(w_new - w_old).max() <= 1e-6 * np.abs(w_old).max()
