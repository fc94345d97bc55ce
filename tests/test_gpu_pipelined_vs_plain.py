"""The pipelined kernels of the headline path against the plain ones through the product path.

k_hstep_rp (role pipeline: loader waves with buffer loads / stores and scalar offsets, progress-slot signalling) and
k_wstats with LDS-DMA loader waves compute exactly what the barrier-phased k_hstep and the synchronously staging
k_wstats compute (same MFMA order per tile), so with every tile in the pipeline (SNMF_HSTEP_SPLIT=0) H after H-only
iterations must agree in every bit and W after full iterations to the summation order of the row sums.  This is the
check that found the two synchronisation bugs and the buffer-store hazard of round 2 (profiles/r02_experiments.md): the
shapes are the ones that showed them -- few row tiles with several tiles per workgroup, short H rows (rp < 256: lanes
past the row duplicate lane 0), a V block whose last cell straddles the end.

Round 3: the tiles of the last PARTIAL round of k_hstep_rp are split by rows over the workgroups that would idle through
it ("the split last round" in csrc/snmf_kernels.h: rp_part_p1 / rp_part_p2 / rp_part_finish).  A split tile sums Lam in k ranges and the numerator in row
parts, i.e. in another fp32 order than the one-workgroup tiles: the default path must equal the plain kernels BIT FOR
BIT on every pipelined tile and to the stated summation-order tolerance (2e-5 relative per element after two
iterations) on the split ones.  The shapes cover 4-way and 2-way splits, no split, and problems smaller than one round
(never split).  SNMF_HSTEP_RP / SNMF_HSTEP_SPLIT / SNMF_WSTATS_NL are read when a plan is created, so all variants
run in this one process.  (src/sparse_nmf.m:189-208 and :215-239 are the updates all variants implement.)
"""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(65, 70, 12000), (129, 70, 12000), (65, 32, 12000), (64, 70, 12000), (97, 96, 20000), (257, 40, 20000),
          (33, 8, 30000), (161, 200, 9000), (257, 256, 30000), (225, 100, 17000), (513, 64, 12000),
          (257, 40, 1000), (65, 70, 3000), (257, 256, 9000), (129, 24, 40),  # fewer tiles than workgroups (never split: latency-bound) / a short partial round
          # 9..16 row tiles: k_hstep_rh (one ratio image, pipelined by half tiles) and k_wstats with loader waves on a
          # compact V image -- the reference's shipped F = 513 at R = 100 / 200 (settings/initial_setting_SNMF_NAT.m:21-29,48-49)
          (513, 100, 12000), (513, 200, 9000), (513, 256, 5000), (385, 100, 12000), (449, 250, 9000), (512, 128, 8000),
          (289, 40, 20000), (513, 200, 100), (513, 100, 9000), (512, 256, 10000), (513, 97, 12000), (512, 99, 9000),
          (513, 194, 12000), (512, 200, 16500),
          # at most two row tiles (a Mel spectrogram: run_basis_train.m:91 on 64 bands): k_hstep_rp with P2 on the B waves of
          # the SIMDs the A team leaves free, k_wstats with consumer TEAMS that take the tiles of a chunk in turn
          (64, 100, 20000), (64, 40, 12000), (32, 100, 12000), (64, 200, 12000), (128, 100, 12000), (64, 100, 9000),
          (64, 128, 70000), (40, 20, 33000), (64, 100, 3000), (32, 8, 100), (64, 256, 9000), (64, 130, 8300),
          # ... whose last tiles are shared by four waves (NK = 7 in pairs, NK = 8 on one row tile, NK = 3 / 5: fewer than four participants / an odd count)
          (64, 200, 37000), (32, 256, 35000), (64, 70, 66000), (64, 160, 68000),
          # one or two column tiles on >= 4 row tiles: k_hstep_rp<., CUT> (every B wave a quarter of the contraction) -- the reference's
          # R = 20 / 10 / 30 settings at F = 513 (settings/bak_IS16_results/initial_setting_SNMF_Techwin_201603_RT.m:47-48)
          (513, 20, 12000), (513, 30, 9000), (513, 10, 9000), (257, 32, 20000), (385, 10, 12000), (257, 64, 9000), (129, 50, 12000),
          # ... and its PAIR form where two column tiles' partials do not fit four ways (16 row tiles; initial_setting_IMCRA.m:47-48 R = 50)
          (513, 50, 12000), (512, 64, 9000), (513, 33, 9000), (481, 40, 12000),
          # ... with THREE or more tiles per workgroup: only then does a loader refill a buffer (LDS-DMA on more than 10 row tiles), and
          # only then did the race between one loader wave's DMA and another's copy-out of the same H rows show (round 5, found by fuzzing)
          (513, 20, 21000), (385, 33, 21157), (422, 32, 17725), (513, 64, 19530), (385, 50, 19546), (421, 30, 17743), (257, 32, 26000)]


SF_SHARED = {(64, 128, 70000), (64, 200, 37000), (32, 256, 35000), (64, 70, 66000), (64, 160, 68000)}  # (on 256 compute units)


def _run(ctx, V, W0, H0, r, *, h_only, iters):
    from se_snmf_nat_amd import Plan
    F, T = V.shape
    kw = dict(w_update_ind=np.zeros(r, bool)) if h_only else {}
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=iters, conv_eps=0.0, cost_check=True, sparsity=1.0, **kw)
    pl.set_v(V); pl.set_w(W0); pl.set_h(H0); pl.init(); pl.run()
    out = (pl.get_h(np.float32), pl.get_w(), pl.describe(), pl.get_objective()[1])
    pl.close()
    return out


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "F%d_r%d_T%d" % s)
def test_pipelined_kernels_equal_plain_kernels(gpu_ctx, shape, monkeypatch):
    F, r, T = shape
    rs = np.random.default_rng(F * 1000 + r)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    for k in ("SNMF_HSTEP_RP", "SNMF_HSTEP_SPLIT", "SNMF_WSTATS_NL"):
        monkeypatch.delenv(k, raising=False)
    h_new, _, geo, obj_new = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    _, w_new, geo_full, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    n_cu = int(re.search(r"n_cu=(\d+)", geo).group(1))
    sr = "k_hstep_sr" in geo  # r <= 64 on 3..16 row tiles: a tile per workgroup cut by row tiles (round 6, csrc/snmf_smallr.h)
    if (T + 31) // 32 <= n_cu and "k_hstep_sf" not in geo:
        # one tile per workgroup: nothing to pipeline, the plan takes the barrier-phased kernels by itself (C1's case)
        assert "k_hstep_rp" not in geo and "k_hstep_rh" not in geo  # (k_wstats keeps its loader waves while a workgroup has several tiles)
        n_full, n_tiles, S = (T + 31) // 32, (T + 31) // 32, 0
    else:
        assert "k_hstep_rp" in geo or "k_hstep_rh" in geo or "k_hstep_sf" in geo or sr  # the pipelined path is what ran
        m = re.search(r"(\d+) of (\d+) tiles pipelined, last round split (\d+) ways", geo)
        if "k_hstep_sf" in geo:  # (F <= 64, r <= 256: a tile per wave; round 6: the tiles of a partial wave level that is the first on its SIMDs are SHARED by four waves)
            assert F <= 64 and r <= 256
            x = int(re.search(r"the last (\d+) shared by four waves each", geo).group(1))
            n_tiles = (T + 31) // 32
            n_full, S = n_tiles - x, (4 if x else 0)
            assert (x > 0) == (shape in SF_SHARED), (shape, x)
        elif sr:  # (never split either; r = 33..64 keep k_hstep_rp<., CUT>: two column tiles do not fit the registers)
            assert r <= 32 and 65 <= F <= 544 and "k_wstats_sr" in geo_full
            n_full, n_tiles, S = (T + 31) // 32, (T + 31) // 32, 0
        else:
            n_full, n_tiles, S = (int(x) for x in m.groups())
    monkeypatch.setenv("SNMF_HSTEP_SPLIT", "0")
    h_ns, _, geo_ns, _ = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    _, w_ns, _, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    assert "split 0 ways" in geo_ns or "k_hstep," in geo_ns or "the last 0 shared" in geo_ns or "k_hstep_sr" in geo_ns
    monkeypatch.setenv("SNMF_HSTEP_RP", "0")
    monkeypatch.setenv("SNMF_WSTATS_NL", "0")
    h_old, _, geo_old, obj_old = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    _, w_old, _, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    assert "k_hstep_rp" not in geo_old and "k_hstep_rh" not in geo_old and "k_hstep_sf" not in geo_old and "k_hstep_sr" not in geo_old and "k_wstats_sr" not in geo_old
    # every tile in the pipeline: the plain kernels bit for bit -- except where k_hstep_rh cuts P2 over the contraction
    # (r = 97..100 on 16 row tiles: four partial sums per numerator instead of one chain), which is a summation order of its own
    lxh = "over the contraction" in geo or sr  # (r = 97..100 four ways, r = 193..200 in wave pairs; k_hstep_sr: eight partial numerators by row tiles, added in wave order)
    if lxh:
        d = np.abs(h_ns - h_old)
        assert (d <= 2e-5 * np.abs(h_old) + 1e-30).all(), d.max()
    else:
        assert np.array_equal(h_ns, h_old)
    assert np.abs(w_ns - w_old).max() <= 1e-6 * np.abs(w_old).max()
    # default path: pipelined tiles bit for bit, split tiles to the summation-order tolerance
    t_split = 32 * n_full if S else T
    if lxh:
        d = np.abs(h_new[:, :t_split] - h_old[:, :t_split])
        assert (d <= 2e-5 * np.abs(h_old[:, :t_split]) + 1e-30).all(), d.max()
    else:
        assert np.array_equal(h_new[:, :t_split], h_old[:, :t_split])
    if S:
        assert n_full < n_tiles
        d = np.abs(h_new[:, t_split:] - h_old[:, t_split:])
        assert (d <= 2e-5 * np.abs(h_old[:, t_split:]) + 1e-30).all(), d.max()
        assert not np.array_equal(h_new[:, t_split:], h_old[:, t_split:]) or T - t_split < 64  # (another order: not the same bits)
    assert np.abs(w_new - w_old).max() <= 2e-6 * np.abs(w_old).max()
    if F in (32, 64) and r <= 128 and (T + 31) // 32 > n_cu:  # (the NK = 4 geometries' KL statistics: k_wstats_teams)
        assert "k_wstats_sf" in geo_full or "consumer teams take the tiles in turn" in geo_full
    for a, b in zip(obj_new, obj_old):
        assert abs(a - b) <= 1e-6 * abs(b)


@pytest.mark.parametrize("shape", [(257, 256, 30000), (256, 230, 17000), (257, 225, 40007), (257, 256, 9000)], ids=lambda s: "F%d_r%d_T%d" % s)
def test_merged_role_h_step_equals_plain(gpu_ctx, shape, monkeypatch):
    """k_hstep_m (csrc/snmf_hstep_m.h: one wave per SIMD runs P1, both epilogues and P2 of its own row / column tiles; opt-in,
    SNMF_HSTEP_M=1) against the barrier-phased k_hstep: the same MFMA order per output tile, so H after H-only iterations
    agrees in every bit, W after full iterations to the summation order of the row sums (src/sparse_nmf.m:189-208)."""
    F, r, T = shape
    rs = np.random.default_rng(F * 1000 + r)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    W0 = rs.random((F, r))
    H0 = rs.random((r, T)).astype(np.float32)
    for k in ("SNMF_HSTEP_RP", "SNMF_HSTEP_SPLIT", "SNMF_WSTATS_NL"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SNMF_HSTEP_M", "1")
    h_m, _, geo, obj_m = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    n_cu = int(re.search(r"n_cu=(\d+)", geo).group(1))
    if "k_hstep_m" not in geo and (T + 31) // 32 > n_cu:
        pytest.skip("k_hstep_m is an experiment: only in -DSNMF_EXPERIMENTS builds (SNMF_EXPERIMENTS=1 python scripts/build_variant.py exp)")
    _, w_m, _, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    assert ("k_hstep_m" in geo) == ((T + 31) // 32 > n_cu)
    monkeypatch.setenv("SNMF_HSTEP_M", "0")
    monkeypatch.setenv("SNMF_HSTEP_RP", "0")
    h_p, _, geo_p, obj_p = _run(gpu_ctx, V, W0, H0, r, h_only=True, iters=2)
    _, w_p, _, _ = _run(gpu_ctx, V, W0, H0, r, h_only=False, iters=3)
    assert "k_hstep_m" not in geo_p and "k_hstep_rp" not in geo_p
    assert np.array_equal(h_m, h_p)
    assert np.abs(w_m - w_p).max() <= 2e-6 * np.abs(w_p).max()
    for a, b in zip(obj_m, obj_p):
        assert abs(a - b) <= 1e-6 * abs(b)
