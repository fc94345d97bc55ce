"""Missing-data-imputation variants: src/snmf_mdi.m, src/snmf_mdi_Sm.m.

CPU: the oracle (oracle/mdi_oracle.py) against properties the reference's formulas imply.
GPU: the device path (snmf_plan_set_mask / get_v_mdi through se_snmf_nat_amd.api.snmf_mdi[_Sm]) against the
oracle: identical early-stop index, H and v_MDI within REL = 1e-4 (Frobenius-relative), every recorded
cost within 1e-5 relative.
"""
import numpy as np
import pytest

from oracle.mdi_oracle import snmf_mdi as oracle_mdi
from oracle.sparse_nmf_oracle import sparse_nmf as oracle_nmf, synth_problem

REL = 1e-4
REL_COST = 1e-5


def problem(F=65, T=90, r=9, miss=0.3, seed=0, soft=False):
    V, W0, H0 = synth_problem(F, T, r)
    rs = np.random.RandomState(seed)
    M = (rs.rand(F, T) > miss).astype(np.float64)
    if soft:
        M = np.clip(M * 0.8 + rs.rand(F, T) * 0.2, 0, 1)
    return V, M, W0, H0


# ---------------------------------------------------------------- CPU: the oracle ------------------
def test_full_mask_reduces_to_the_plain_solver():
    V, M, W0, H0 = problem()
    p = dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=1e-4, max_iter=40, init_w=W0, init_h=H0, cost_check=1)
    v_mdi, h, o = oracle_mdi(V, np.ones_like(V), p)
    w2, h2, o2 = oracle_nmf(V, dict(cf="kl", sparsity=0.5, conv_eps=1e-4, max_iter=40, init_w=W0, init_h=H0, cost_check=1))
    assert o["n_iter"] == o2["n_iter"]
    np.testing.assert_allclose(h, h2, rtol=1e-12)
    np.testing.assert_allclose(o["cost"], o2["cost"], rtol=1e-12)
    np.testing.assert_allclose(v_mdi, np.maximum(V, 1e-9), rtol=1e-12)  # nothing to impute


def test_observed_entries_survive_and_missing_ones_follow_the_model():
    V, M, W0, H0 = problem()
    p = dict(cf="kl", sparsity_mdi=0.1, conv_eps_mdi=0, max_iter=60, init_w=W0, init_h=H0, cost_check=1)
    v_mdi, h, o = oracle_mdi(V, M, p)
    obs = M == 1
    np.testing.assert_allclose(v_mdi[obs], np.maximum(V, 1e-9)[obs], rtol=1e-12)
    lam = np.maximum(o["w"] @ h, 1e-9)
    Nt = (np.maximum(V, 1e-9) * M).sum(0) / np.maximum((lam * M).sum(0), 1e-9)
    np.testing.assert_allclose(v_mdi[~obs], (Nt[None, :] * lam)[~obs], rtol=1e-10)  # :302-305
    assert np.all(np.diff(o["cost"][1:]) <= 1e-9 * o["cost"][1])  # imputed entries sit on the model: cost keeps falling
    # the imputation recovers the low-rank truth far better than the masked start
    err = np.linalg.norm((v_mdi - V)[~obs]) / np.linalg.norm(V[~obs])
    assert err < 0.5


def test_parameter_quirks():
    V, M, W0, H0 = problem(F=20, T=12, r=3)
    with pytest.raises(KeyError):  # p.sparsity present, p.sparsity_mdi absent: missing field (:87-89, :150)
        oracle_mdi(V, M, dict(sparsity=1, conv_eps_mdi=0, init_w=W0, init_h=H0, cost_check=1))
    with pytest.raises(KeyError):  # :270
        oracle_mdi(V, M, dict(init_w=W0, init_h=H0))
    v1, _, o1 = oracle_mdi(V, M, dict(init_w=W0, init_h=H0, cost_check=1, max_iter=5))  # defaults 0 / 0
    assert o1["n_iter"] == 5 and len(o1["cost"]) == 5


# ---------------------------------------------------------------- GPU ------------------------------
CASES = [
    dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=0, max_iter=25),
    dict(cf="kl", sparsity_mdi=5.0, conv_eps_mdi=1e-3, max_iter=80),                       # early stop
    dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=0, max_iter=20, soft=True),               # snmf_mdi_Sm
    dict(cf="ed", sparsity_mdi=0.05, conv_eps_mdi=0, max_iter=20),
    dict(cf="is", sparsity_mdi=0.01, conv_eps_mdi=0, max_iter=15),
    dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=1e-4, max_iter=60, h_only=True),          # supervised imputation
    dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=0, max_iter=12, cost_check=0),
    dict(cf="kl", sparsity_mdi=0.5, conv_eps_mdi=1e-4, max_iter=40, w_only=True),         # dictionary re-trained on gappy data
    dict(cf="ed", sparsity_mdi=0.05, conv_eps_mdi=0, max_iter=10, w_only=True, cost_check=0),
    dict(cf="kl", sparsity_mdi=1.0, conv_eps_mdi=0, max_iter=15, F=257, T=1000, r=40),     # extra-row mode, many tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}={c[k]}" for k in c if k not in ("sparsity_mdi",)))
def test_device_mdi_matches_the_oracle(gpu_ctx, case):
    from se_snmf_nat_amd import snmf_mdi, snmf_mdi_Sm
    c = dict(case)
    soft, h_only, w_only = c.pop("soft", False), c.pop("h_only", False), c.pop("w_only", False)
    V, M, W0, H0 = problem(c.pop("F", 65), c.pop("T", 90), c.pop("r", 9), soft=soft)
    p = dict(c, init_w=W0, init_h=H0)
    p.setdefault("cost_check", 1)
    if h_only:
        p["w_update_ind"] = np.zeros(W0.shape[1], bool)
    if w_only:
        p["h_update_ind"] = np.zeros(W0.shape[1], bool)
    v_ref, h_ref, o_ref = oracle_mdi(V, M, p)
    v_dev, h_dev, o_dev = (snmf_mdi_Sm if soft else snmf_mdi)(V, M, p, ctx=gpu_ctx)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert o_dev["n_iter"] == o_ref["n_iter"]
    assert rel(h_dev, h_ref) < REL
    assert rel(v_dev, v_ref) < REL
    if p["cost_check"]:
        assert len(o_dev["cost"]) == len(o_ref["cost"])
        np.testing.assert_allclose(o_dev["cost"], o_ref["cost"], rtol=REL_COST, atol=2e-7 * V.sum())
    obs = M == 1
    if not soft:  # observed entries are the input's, bit for bit in fp32
        assert np.array_equal(v_dev[obs].astype(np.float32), np.maximum(V, 1e-9)[obs].astype(np.float32))


@pytest.mark.gpu
def test_mdi_state_rules(gpu_ctx):
    from se_snmf_nat_amd import Plan, SnmfError
    V, M, W0, H0 = problem(F=33, T=40, r=5)
    with pytest.raises(SnmfError):  # nothing to update: not an imputation problem
        Plan(gpu_ctx, 33, 40, 5, h_update_ind=np.zeros(5, bool), w_update_ind=np.zeros(5, bool)).set_mask(M)
    pl = Plan(gpu_ctx, 33, 40, 5, max_iter=5, cost_check=True, sparsity=0.1)
    pl.set_mask(M); pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
    v1 = pl.get_v_mdi()
    with pytest.raises(SnmfError):  # V is state: a second solve needs set_v again
        pl.init()
    pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
    assert np.array_equal(v1, pl.get_v_mdi())  # reproducible
    pl.close()
