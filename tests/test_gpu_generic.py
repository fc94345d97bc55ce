"""Shapes beyond the fused kernels' LDS / register envelope (F + r > ~2540; W updates with r > 1024 under KL or
F = 32n+1): src/sparse_nmf.m accepts any size, so the plan runs the same iteration with its intermediates in HBM
(csrc/snmf_generic.h) instead of refusing.  Parity against the fp64 oracle at north_star's tolerance (1e-4 on W and H,
1e-5 on the objective), through the product path; every case checks that the out-of-envelope path is what ran.
(src/sparse_nmf.m:186-286 is what both implement.)
"""
import numpy as np
import pytest

from oracle.sparse_nmf_oracle import sparse_nmf as onmf

pytestmark = pytest.mark.gpu

CASES = {
    "kl_full_F2700": dict(F=2700, T=700, r=40, p=dict(cf="kl", sparsity=5, max_iter=8, conv_eps=0)),
    "kl_full_F2049_r600": dict(F=2049, T=400, r=600, p=dict(cf="kl", sparsity=1, max_iter=5, conv_eps=0)),
    "ed_full_F2600": dict(F=2600, T=500, r=60, p=dict(cf="ed", sparsity=5, max_iter=8, conv_eps=0)),
    "is_honly_F2800": dict(F=2800, T=300, r=30, p=dict(cf="is", sparsity=0.1, max_iter=8, conv_eps=0), mode="h"),
    "b15_full_F2600": dict(F=2600, T=300, r=24, p=dict(cf="x", beta=1.5, sparsity=1, max_iter=6, conv_eps=0)),
    "kl_full_r1100": dict(F=200, T=600, r=1100, p=dict(cf="kl", sparsity=2, max_iter=5, conv_eps=0)),
    "kl_wonly_r1100_splits": dict(F=129, T=5000, r=1100, p=dict(cf="kl", sparsity=2, max_iter=4, conv_eps=0), mode="w"),
    "ed_wonly_r1100_xr": dict(F=129, T=2500, r=1100, p=dict(cf="ed", sparsity=2, max_iter=4, conv_eps=0), mode="w"),
    "kl_semi_early_stop_matrix_sparsity": dict(F=2600, T=260, r=20, p=dict(cf="kl", max_iter=60, conv_eps=1e-2), mode="semi",
                                               sp="mat"),
    "kl_nocost_F2600": dict(F=2600, T=200, r=16, p=dict(cf="kl", sparsity=5, max_iter=6, conv_eps=0, cost_check=0)),
}


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("name", list(CASES))
def test_out_of_envelope_shapes_match_the_oracle(gpu_ctx, name):
    from se_snmf_nat_amd import Plan, sparse_nmf
    c = CASES[name]
    F, T, r = c["F"], c["T"], c["r"]
    rs = np.random.default_rng(F + T + r)
    V = rs.gamma(0.5, 1.0, (F, 10)) @ rs.gamma(0.3, 1.0, (10, T)) + 1e-3
    p = dict(c["p"], init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    p.setdefault("cost_check", 1)
    if c.get("sp") == "mat":
        p["sparsity"] = rs.random((r, T)) * 3
    mode = c.get("mode", "full")
    if mode == "h":
        p["w_update_ind"] = np.zeros(r, bool)
    elif mode == "w":
        p["h_update_ind"] = np.zeros(r, bool)
    elif mode == "semi":
        p["w_update_ind"] = np.arange(r) >= r // 2
    pl = Plan(gpu_ctx, F, T, r, beta={"kl": 1.0, "ed": 2.0, "is": 0.0}.get(p["cf"], p.get("beta", 1.0)), max_iter=2,
              conv_eps=0.0, cost_check=True, sparsity=1.0,
              **({"w_update_ind": p["w_update_ind"]} if "w_update_ind" in p else {}),
              **({"h_update_ind": p["h_update_ind"]} if "h_update_ind" in p else {}))
    assert "out-of-envelope path" in pl.describe()
    pl.close()
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
    wr, hr, orf = onmf(V, p)
    assert o["n_iter"] == orf["n_iter"]
    if name.startswith("kl_semi_early"):
        assert 1 < o["n_iter"] < 60
    assert np.isfinite(w).all() and np.isfinite(h).all()
    assert rel(w, wr) < 1e-4 and rel(h, hr) < 1e-4
    if p["cost_check"]:
        n = len(orf["cost"])
        assert np.max(np.abs(o["cost"][:n] - orf["cost"][:n]) / np.abs(orf["cost"][:n])) < 1e-5
        assert np.max(np.abs(o["div"][:n] - orf["div"][:n]) / np.abs(orf["div"][:n])) < 1e-5


def test_mdi_still_refuses_out_of_envelope_shapes(gpu_ctx):
    from se_snmf_nat_amd import SnmfError, snmf_mdi
    rs = np.random.default_rng(0)
    V = rs.random((2700, 64)) + 0.1
    M = (rs.random((2700, 64)) > 0.2).astype(float)
    p = dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=0, max_iter=3, cost_check=1, init_w=rs.random((2700, 8)), init_h=rs.random((8, 64)))
    with pytest.raises(SnmfError, match="too large"):
        snmf_mdi(V, M, p, ctx=gpu_ctx)


def test_out_of_envelope_shape_over_a_device_list(gpu_ctx):
    """The multi-rank entry shards the frames of an out-of-envelope problem like any other: the statistics buffer and the
    exchange are the fast path's (csrc/snmf_multi.h), only the per-rank products differ."""
    from se_snmf_nat_amd import sparse_nmf
    rs = np.random.default_rng(9)
    F, T, r = 2600, 360, 24
    V = rs.gamma(0.5, 1.0, (F, 10)) @ rs.gamma(0.3, 1.0, (10, T)) + 1e-3
    p = dict(cf="kl", sparsity=2, max_iter=6, conv_eps=0, cost_check=1, init_w=rs.random((F, r)), init_h=rs.random((r, T)))
    w1, h1, o1 = sparse_nmf(V, p, ctx=gpu_ctx)
    w2, h2, o2 = sparse_nmf(V, p, devices=[0, 0])
    assert rel(w2, w1) < 1e-5 and rel(h2, h1) < 1e-5
    assert np.max(np.abs(o2["cost"] - o1["cost"]) / np.abs(o1["cost"])) < 1e-6
