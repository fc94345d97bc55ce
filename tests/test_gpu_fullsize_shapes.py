"""The full-size shapes of scripts/bench_f513.py -- eight or more tiles per workgroup, which is where a
loader refills a tile buffer and a workgroup's hand-off buffers wrap around -- against the fp64 oracle on the same inputs:
three iterations each, W and H at 1e-4, costs at 1e-5 (the tolerances of tests/test_gpu_parity.py).  The unit-test shapes are
sized for seconds and have two or three tiles per workgroup; a race in a refill path (profiles/r05_experiments.md section 10) was
invisible to them."""
import numpy as np
import pytest

from oracle.sparse_nmf_oracle import sparse_nmf as oracle_nmf

pytestmark = pytest.mark.gpu

SHAPES = [  # name, F, T, r, mode, kernel the plan must report
    ("R20_513", 513, 72000, 20, "full", "k_hstep_sr"),         # settings/bak_IS16_results/initial_setting_SNMF_Techwin_201603_RT.m:47-48
    ("R30_513_h", 513, 72000, 30, "h", "k_hstep_sr"),
    ("R10_513_w", 513, 72000, 10, "w", "k_wstats_sr"),         # ... R_d = 10, the W-only solve of run_basis_DNMF.m:53 at those settings
    ("R50_513", 513, 72000, 50, "full", "wave pairs cut"),     # settings/bak_IS16_results/initial_setting_IMCRA.m:47-48
    ("R32_257", 257, 100000, 32, "full", "k_hstep_sr"),
    ("R64_257", 257, 100000, 64, "full", "over the contraction"),
    ("mel_R100", 64, 72000, 100, "full", "k_iter_sf"),          # run_basis_train.m:90-91
    # ... and the reference's shipped geometry on the kernels of rounds 3 / 4 (k_hstep_rh in both cut modes, k_wstats with LX columns)
    ("a11", 513, 72000, 100, "full", "k_hstep_rh"),             # run_basis_train.m:88
    ("c4_solve1", 513, 100000, 200, "h", "k_hstep_rh"),         # run_basis_DNMF.m:40
    ("c4_solve2", 513, 100000, 100, "w", "wstats: NK=4"),       # run_basis_DNMF.m:47
    ("mel_dnmf_h", 64, 100000, 200, "h", "k_hstep_sf"),         # run_basis_DNMF_Mel.m:75
    # BASELINE configs[4] (beta = 2, lambda = 50, r = 512) at 60000 frames -- seven tiles per workgroup; the oracle needs ~10 s for these
    ("c5_60k", 513, 60000, 512, "ed", "Gram matrix"),
]


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("name,F,T,r,mode,kern", SHAPES, ids=[s[0] for s in SHAPES])
def test_bench_shapes_at_full_size_against_the_oracle(gpu_ctx, name, F, T, r, mode, kern):
    from se_snmf_nat_amd import Plan, sparse_nmf
    rs = np.random.default_rng(F + r)
    V = rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3
    W0, H0 = rs.random((F, r)), rs.random((r, T))
    p = dict(cf="kl", sparsity=5.0, max_iter=3, conv_eps=0, cost_check=1, init_w=W0, init_h=H0)
    kw = {}
    beta = 1.0
    if mode == "ed":
        p.update(cf="ed", sparsity=50.0)
        beta = 2.0
    if mode == "h":
        p["w_update_ind"] = np.zeros(r, bool)
        kw["w_update_ind"] = np.zeros(r, bool)
    if mode == "w":
        p["h_update_ind"] = np.zeros(r, bool)
        kw["h_update_ind"] = np.zeros(r, bool)
    pl = Plan(gpu_ctx, F, T, r, beta=beta, max_iter=3, conv_eps=0.0, cost_check=True, sparsity=p["sparsity"], **kw)
    geo = pl.describe()
    pl.close()
    assert kern in geo, geo  # the kernel this test is about is the one the plan takes
    w, h, o = sparse_nmf(V, p, ctx=gpu_ctx)
    wr, hr, orf = oracle_nmf(V, p)
    assert rel(w, wr) < 1e-4 and rel(h, hr) < 1e-4, (rel(w, wr), rel(h, hr))
    np.testing.assert_allclose(o["cost"], orf["cost"], rtol=1e-5)
