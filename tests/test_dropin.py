"""The device-resident callers (include/snmf.h: snmf_run_basis_dnmf_*, snmf_run_basis_train_audio_f64) and the chunked
host <-> device pipeline behind every host-array entry (csrc/snmf_tu_xfer.hip).

The resident entries must give the SAME BITS as the three separate sparse_nmf calls of run_basis_DNMF.m:36-55 through the
one-shot C entry -- they only remove host round trips -- and the golden loop / the oracle chain within the solver tolerances.
"""
import os
import threading

import numpy as np
import pytest

from test_oracle import GOLD

REL_WH = 1e-4


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_philox_known_answers():
    """Philox-4x32-10 (the generator of snmf_plan_set_h_random) against the known-answer vectors published with the
    algorithm (Random123 kat_vectors: zero, all-ones and pi-digit counters / keys)."""
    from se_snmf_nat_amd.api import philox4x32_10, philox_uniform
    kat = [([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, exp in kat:
        out = philox4x32_10([np.array([c], dtype=np.uint64) for c in ctr], *key)
        assert [int(o[0]) for o in out] == exp
    u = philox_uniform(7, 13, 1001)
    assert u.shape == (13, 1001) and u.dtype == np.float32 and 0.0 < u.min() and u.max() < 1.0
    assert abs(float(u.mean()) - 0.5) < 0.01
    assert not np.array_equal(u, philox_uniform(8, 13, 1001))


def _dnmf_problem(F=513, T=700, R_x=20, R_d=12, seed=3):
    rs = np.random.default_rng(seed)
    X = rs.gamma(0.5, 1.0, (F, 9)) @ rs.gamma(0.3, 1.0, (9, T)) + 1e-9
    D = rs.gamma(0.5, 1.0, (F, 7)) @ rs.gamma(0.3, 1.0, (7, T)) + 1e-9
    return X + D, X, D, rs.random((F, R_x + R_d)) + 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("conv_eps", [0.0, 1e-3])
def test_resident_dnmf_equals_three_calls_bit_for_bit(gpu_ctx, dtype, conv_eps):
    """snmf_run_basis_dnmf_* against three snmf_sparse_nmf_* calls (A_hat through the host in between): same bits in
    B_hat and A_hat, with and without early stops; non-tight leading dimensions on the way in."""
    from se_snmf_nat_amd import run_basis_dnmf
    Y, X, D, B = _dnmf_problem()
    p = dict(cf="kl", sparsity=5, max_iter=25, conv_eps=conv_eps, cost_check=1, random_seed=1)
    # a view with a leading dimension larger than F (columns of a taller Fortran array)
    tall = np.asfortranarray(np.zeros((Y.shape[0] + 5, Y.shape[1]), dtype))
    tall[:Y.shape[0]] = Y
    Yv = tall[:Y.shape[0]]
    B1, A1 = run_basis_dnmf(Yv, X, D, B, 20, 12, p, ctx=gpu_ctx, dtype=dtype, resident=True)
    B3, A3 = run_basis_dnmf(Y, X, D, B, 20, 12, p, ctx=gpu_ctx, dtype=dtype, resident=False)
    assert B1.dtype == np.dtype(dtype) and B1.shape == B3.shape and A1.shape == A3.shape
    assert np.array_equal(B1, B3) and np.array_equal(A1, A3)
    np.testing.assert_allclose(np.sqrt((B1.astype(np.float64) ** 2).sum(0)), 1.0, rtol=1e-6)


@pytest.mark.gpu
def test_resident_dnmf_against_the_golden_loop(gpu_ctx):
    """The committed oracle run of run_basis_DNMF.m:36-55 (tests/golden/dnmf_loop_513x64_r20_20.npz) through the resident entry."""
    from se_snmf_nat_amd import run_basis_dnmf
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    d = dict(np.load(os.path.join(GOLD, "dnmf_loop_513x64_r20_20.npz")))
    Y = ref["Y"]
    X = (Y * d["mask"] + 1e-9).astype(np.float32)
    D = (Y - X + 2e-9).astype(np.float32)
    Bs = np.concatenate([ref["B"][:, :20], ref["B"][:, 100:120]], axis=1)
    p = dict(cf="kl", sparsity=5, max_iter=30, conv_eps=1e-3, cost_check=1)
    B_hat, A_hat = run_basis_dnmf(Y, X, D, Bs, 20, 20, p, ctx=gpu_ctx, resident=True)
    assert rel(B_hat, d["B_hat"]) < REL_WH and rel(A_hat, d["A_hat"]) < REL_WH


@pytest.mark.gpu
def test_initial_h_drawn_on_the_device(gpu_ctx):
    """snmf_plan_set_h_random writes philox_uniform(seed, r, T) into the resident H (bit for bit; nothing crosses PCIe), and the
    resident loop with h0="device" equals the three-call path started from those numbers."""
    from se_snmf_nat_amd import Plan, run_basis_dnmf
    from se_snmf_nat_amd.api import philox_uniform
    plan = Plan(gpu_ctx, 40, 3001, 37, max_iter=1, sparsity=1.0, cost_check=True)
    plan.set_h_random(12345678901)
    got = plan.get_h(np.float32)
    plan.close()
    assert np.array_equal(got, philox_uniform(12345678901, 37, 3001))
    Y, X, D, B = _dnmf_problem(F=257, T=400, R_x=8, R_d=8)
    p = dict(cf="kl", sparsity=5, max_iter=10, conv_eps=0, cost_check=1, random_seed=4)
    B1, A1 = run_basis_dnmf(Y, X, D, B, 8, 8, p, ctx=gpu_ctx, resident=True, h0="device")
    B3, A3 = run_basis_dnmf(Y, X, D, B, 8, 8, p, ctx=gpu_ctx, resident=False, h0="device")
    assert np.array_equal(B1, B3) and np.array_equal(A1, A3)


@pytest.mark.gpu
def test_resident_dnmf_errors(gpu_ctx):
    """Errors of the resident entry: R_x + R_d must be the rank; a non-scalar sparsity is the dimension error of the reference's
    W-only solves (src/sparse_nmf.m:192 with R_x rows of H against R_x + R_d rows of p.sparsity)."""
    from se_snmf_nat_amd import run_basis_dnmf
    from se_snmf_nat_amd.api import SnmfError
    Y, X, D, B = _dnmf_problem(F=65, T=100, R_x=4, R_d=4)
    p = dict(cf="kl", sparsity=5, max_iter=3, conv_eps=0, cost_check=1)
    with pytest.raises(SnmfError):
        run_basis_dnmf(Y, X, D, B[:, :7], 4, 4, p, ctx=gpu_ctx)
    with pytest.raises(SnmfError):
        run_basis_dnmf(Y, X, D, B, 4, 4, dict(p, sparsity=np.ones(8)), ctx=gpu_ctx)
    with pytest.raises(SnmfError):
        q = dict(p)
        q.pop("cost_check")
        run_basis_dnmf(Y, X, D, B, 4, 4, q, ctx=gpu_ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_chunked_transfers_round_trip(gpu_ctx, dtype):
    """A matrix of several transfer chunks (16 MiB each) with a non-tight leading dimension goes host -> device -> host through
    the pinned pipeline and comes back as its fp32 rounding, element for element; the context's counters see the bytes."""
    from se_snmf_nat_amd import Plan
    r, T = 200, 70_001  # 56 MB as fp32: four chunks, the last one ragged
    rs = np.random.default_rng(11)
    big = np.asfortranarray(rs.random((r + 3, T)).astype(dtype))
    H = big[:r]  # leading dimension r + 3
    plan = Plan(gpu_ctx, 33, T, r, max_iter=1, sparsity=0.0, cost_check=True)
    gpu_ctx.xfer_stats(reset=True)
    plan.set_h(H)
    out = np.asfortranarray(np.full((r + 2, T), -1.0, dtype))
    import ctypes as C
    from se_snmf_nat_amd import _lib
    fn = plan._lib.snmf_plan_get_h_f64 if dtype == np.float64 else plan._lib.snmf_plan_get_h_f32
    _lib.check(fn(plan._h, C.c_void_p(out.ctypes.data), r + 2, 0))
    st = gpu_ctx.xfer_stats()
    plan.close()
    assert np.array_equal(out[:r], H.astype(np.float32).astype(dtype))
    assert (out[r:] == -1.0).all()  # rows beyond r of the caller's array are not touched
    assert st["h2d_bytes"] == r * T * np.dtype(dtype).itemsize and st["d2h_bytes"] == r * T * np.dtype(dtype).itemsize
    assert st["h2d_calls"] == 1 and st["d2h_calls"] == 1 and st["h2d_wall_s"] > 0 and st["d2h_wall_s"] > 0


@pytest.mark.gpu
def test_transfers_from_two_host_threads(lib):
    """Two contexts fed from two host threads at once (what the multi-device entry and the resident DNMF loop do): the worker
    pool serves both callers, each pipeline keeps its own bounce buffers, nothing is mixed up."""
    from se_snmf_nat_amd import Context, Plan
    r, T = 96, 60_000
    res = {}

    def work(tag):
        ctx = Context(0)
        rs = np.random.default_rng(tag)
        H = np.asfortranarray(rs.random((r, T)))
        plan = Plan(ctx, 33, T, r, max_iter=1, sparsity=0.0, cost_check=True)
        for _ in range(3):
            plan.set_h(H)
            got = plan.get_h(np.float64)
        res[tag] = np.array_equal(got, H.astype(np.float32).astype(np.float64))
        plan.close()
        ctx.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert res == {1: True, 2: True}


@pytest.mark.gpu
def test_audio_entries_equal_the_feature_path(gpu_ctx):
    """run_basis_DNMF(x, d, B, p) through ONE call (audio in, B_hat out: snmf_run_basis_dnmf_audio_f64) against the same loop
    fed with features that went to the host and back (three stft calls + the resident feature entry): the feature sets are
    produced by the same kernels, so B_hat agrees to fp32 rounding of y = x + d."""
    from se_snmf_nat_amd import run_basis_dnmf, train
    import oracle.frontend_oracle as fo
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"].astype(np.float64)
    x, d = s[:9000], s[9000:19000][::-1].copy()
    p = dict(fo.default_params(), cf="kl", sparsity=5, max_iter=12, conv_eps=1e-3, cost_check=1, random_seed=1, R_x=10, R_d=12)
    B = np.random.RandomState(5).rand(513, 22) + 0.05
    one = train.run_basis_DNMF(x, d, B, p, ctx=gpu_ctx)
    Y, X, D = train._dnmf_features(x, d, p, gpu_ctx)
    three, _ = run_basis_dnmf(Y, X, D, B, 10, 12, p, ctx=gpu_ctx, dtype=np.float32)
    assert rel(one, three) < 1e-6
    dev = train.run_basis_DNMF(x, d, B, p, ctx=gpu_ctx, h0="device")  # initial activations drawn on the device: another start
    assert dev.shape == one.shape and np.isfinite(dev).all() and rel(dev, one) > 1e-6
    np.testing.assert_allclose(np.sqrt((dev ** 2).sum(0)), 1.0, rtol=1e-6)


@pytest.mark.gpu
def test_training_entry_in_exemplar_mode(gpu_ctx):
    """run_basis_train.m:84,:95-96 with p.train_Exemplar: no solve, the dictionaries are the (normalised) exemplar columns of
    TF_mag / TF_Mel and the activations the scalar 0 -- features and the gather on the device, against the oracle's features."""
    from se_snmf_nat_amd import train
    import oracle.frontend_oracle as fo
    s = np.load(os.path.join(GOLD, "frontend_audio.npz"))["samples"].astype(np.float64)
    p = dict(fo.default_params(), cf="kl", sparsity=5, max_iter=20, conv_eps=0, cost_check=1, cluster_buff=1, train_Exemplar=1)
    idx = np.random.RandomState(6).choice(114, size=9, replace=False) + 1
    out = train.run_basis_train_signal(s, 9, p, sample_idx=idx, ctx=gpu_ctx)
    TF = fo.dft_features(s, p)
    TM = fo.mel_features(TF, p)
    for key, M in (("B_DFT_sub", TF), ("B_Mel_sub", TM)):
        ref = M[:, idx - 1]
        ref = ref / np.sqrt((ref ** 2).sum(0)) + 1e-9
        assert rel(out[key], ref) < 1e-5, key
    assert out["A_DFT_sub"] == 0 and out["A_Mel_sub"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 3])
@pytest.mark.parametrize("conv_eps", [0.0, 1e-3])
def test_resident_dnmf_over_a_device_list_equals_the_single_device_call(gpu_ctx, n_ranks, conv_eps):
    """snmf_run_basis_dnmf_multi_f64 (run_basis_DNMF.m:36-55 with the frames of all three solves sharded over a device list, every
    rank's A_hat columns resident between the solves; csrc/snmf_multi.h) against the single-device resident call: frames are
    independent given W, so A_hat -- nothing but the two cost scalars is exchanged in solve 1 -- comes out BIT FOR BIT, early
    stops included; B_hat's statistics are summed per rank and then across the ranks (fp64), which moves it by a few fp32 ulp.
    The ranks share the one device of the test box (EVENTS ordering)."""
    from se_snmf_nat_amd import run_basis_dnmf
    Y, X, D, B = _dnmf_problem(F=513, T=900, R_x=12, R_d=9, seed=8)
    p = dict(cf="kl", sparsity=5, max_iter=40, conv_eps=conv_eps, cost_check=1, random_seed=1)
    B1, A1 = run_basis_dnmf(Y, X, D, B, 12, 9, p, ctx=gpu_ctx)
    Bm, Am = run_basis_dnmf(Y, X, D, B, 12, 9, p, devices=[0] * n_ranks)
    assert np.array_equal(Am, A1)
    assert rel(Bm, B1) < 5e-6
    # a second call reuses the device list's team (contexts, gather buffers): the same bits again
    Bm2, Am2 = run_basis_dnmf(Y, X, D, B, 12, 9, p, devices=[0] * n_ranks)
    assert np.array_equal(Bm2, Bm) and np.array_equal(Am2, Am)
    # ... and the three-call path over the same device list (round 4's form) agrees with the resident one bit for bit
    B3, A3 = run_basis_dnmf(Y, X, D, B, 12, 9, p, devices=[0] * n_ranks, resident=False)
    assert np.array_equal(B3, Bm) and np.array_equal(A3, Am)


@pytest.mark.gpu
def test_resident_multi_dnmf_golden_loop_and_device_rng(gpu_ctx):
    """The committed oracle run of run_basis_DNMF.m:36-55 through the device-list entry (two ranks), and the device generator:
    rank g draws ITS columns of the (R_x + R_d) x T Philox draw, so h0 = "device" gives the single-device A_hat bit for bit."""
    from se_snmf_nat_amd import run_basis_dnmf
    ref = dict(np.load(os.path.join(GOLD, "ref_data.npz")))
    d = dict(np.load(os.path.join(GOLD, "dnmf_loop_513x64_r20_20.npz")))
    Y = ref["Y"]
    X = (Y * d["mask"] + 1e-9).astype(np.float32)
    D = (Y - X + 2e-9).astype(np.float32)
    Bs = np.concatenate([ref["B"][:, :20], ref["B"][:, 100:120]], axis=1)
    p = dict(cf="kl", sparsity=5, max_iter=30, conv_eps=1e-3, cost_check=1)
    B_hat, A_hat = run_basis_dnmf(Y, X, D, Bs, 20, 20, p, devices=[0, 0])
    assert rel(B_hat, d["B_hat"]) < REL_WH and rel(A_hat, d["A_hat"]) < REL_WH
    Y2, X2, D2, B2 = _dnmf_problem(F=257, T=403, R_x=8, R_d=7)
    q = dict(cf="kl", sparsity=5, max_iter=10, conv_eps=0, cost_check=1, random_seed=4)
    B1, A1 = run_basis_dnmf(Y2, X2, D2, B2, 8, 7, q, ctx=gpu_ctx, h0="device")
    Bm, Am = run_basis_dnmf(Y2, X2, D2, B2, 8, 7, q, devices=[0, 0, 0], h0="device")
    assert np.array_equal(Am, A1) and rel(Bm, B1) < 5e-6
