#!/usr/bin/env python3
"""What a caller of the DROP-IN boundary waits for: host arrays in, arrays out (the reference hands over
MATLAB doubles), split into host->device / solve / device->host, next to the bare pinned hipMemcpy rate of the same box.

  a11   run_basis_train.m:88 at the shipped settings: sparse_nmf(513 x 72000 fp64, r = 100, KL, full update, 100 iterations)
  c2    BASELINE configs[1]: sparse_nmf(257 x 100000 fp64, r = 256, KL, 200 iterations)
  c4    BASELINE configs[3] on one GPU: run_basis_DNMF's 3 solves x 50 iterations, 513 x 100000, R_x = R_d = 100 --
        as three sparse_nmf calls (round 3's path), as ONE resident call on features (fp64 and fp32 host arrays), and from the
        two waveforms (audio in, B_hat out)
  mel   run_basis_train.m:91: the Mel solve (64 x 72000, r = 100) -- the HBM-side shape
  c4mel the three solves of run_basis_DNMF_Mel.m (64 Mel bands, 100000 frames): one resident call, and the solves alone
  c4m   BASELINE configs[3] behind a DEVICE LIST (snmf_run_basis_dnmf_multi_f64): n = 2, 4, 8 ranks that share device 0 (the box
        has one GPU: EVENTS ordering), the whole 513 x 100000 problem sharded; beside it n x (one shard's three resident solves)

All random draws and host-side array preparation happen OUTSIDE the timed regions.  One JSON line per measurement:
  python scripts/bench_dropin.py [pcie a11 c2 c4 mel] > profiles/r04_dropin.jsonl
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from se_snmf_nat_amd import Context, Plan, run_basis_dnmf, sparse_nmf  # noqa: E402

which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["pcie", "a11", "c2", "c4"]
ctx = Context(0)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def synth(F, T, r, seed=0, dtype=np.float64):
    rd = np.random.default_rng(seed)
    Wt = rd.gamma(0.5, 1.0, size=(F, r)).astype(np.float32)
    Ht = rd.gamma(0.3, 1.0, size=(r, T)).astype(np.float32)
    V = np.asfortranarray(Wt @ Ht + 1e-9, dtype=dtype)
    return V, np.asfortranarray(rd.random((F, r)), dtype=dtype), np.asfortranarray(rd.random((r, T)), dtype=dtype)


def resident_solve_s(F, T, r, iters, V, W0, H0, **kw):
    """the same solve with the data already in HBM: the kernels' share of a drop-in call"""
    plan = Plan(ctx, F, T, r, max_iter=iters, conv_eps=0.0, cost_check=True, **kw)
    plan.set_v(V); plan.set_w(W0); plan.set_h(H0)
    best = 1e9
    for _ in range(2):
        plan.init(); ctx.sync()
        t = time.perf_counter(); plan.run(); ctx.sync()
        best = min(best, time.perf_counter() - t)
    plan.close()
    return best


def one_shot(name, F, T, r, iters, dtype, **pk):
    V, W0, H0 = synth(F, T, r, dtype=dtype)
    p = dict(cf="kl", sparsity=5, max_iter=iters, conv_eps=0, cost_check=1, init_w=W0, init_h=H0, **pk)
    sparse_nmf(V[:, :2048], dict(p, init_h=H0[:, :2048]), ctx=ctx, dtype=dtype)  # code objects loaded, transfer buffers pinned
    best = None
    for _ in range(2):
        ctx.xfer_stats(reset=True)
        t = time.perf_counter()
        w, h, o = sparse_nmf(V, p, ctx=ctx, dtype=dtype)
        dt = time.perf_counter() - t
        st = ctx.xfer_stats()
        if best is None or dt < best[0]:
            best = (dt, st)
    dt, st = best
    solve = resident_solve_s(F, T, r, iters, V, W0, H0, beta=1.0, sparsity=5.0)
    emit(config=name, host_dtype=np.dtype(dtype).name, call_s=dt, solver_iterations_per_s=iters / dt,
         h2d_s=st["h2d_wall_s"], h2d_GBps=st["h2d_bytes"] / max(st["h2d_wall_s"], 1e-12) / 1e9, h2d_host_copy_s=st["h2d_host_copy_s"],
         d2h_s=st["d2h_wall_s"], d2h_GBps=st["d2h_bytes"] / max(st["d2h_wall_s"], 1e-12) / 1e9, d2h_host_copy_s=st["d2h_host_copy_s"],
         resident_solve_s=solve, other_s=dt - st["h2d_wall_s"] - st["d2h_wall_s"] - solve,
         call_over_resident=dt / solve, h2d_MB=st["h2d_bytes"] / 1e6, d2h_MB=st["d2h_bytes"] / 1e6,
         note="other_s = plan creation / destruction, the mirror's copies of the in/out arrays (what mxDuplicateArray is in the MEX "
              "shim), minus the tail of the upload pipeline that runs under the solve")


if "pcie" in which:
    # bare pinned hipMemcpy on this box, both directions (torch's pinned allocator + copy_ = hipMemcpyAsync on pinned memory)
    import torch
    n = 512 << 20
    hp = torch.empty(n, dtype=torch.uint8).pin_memory()
    dv = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = {}
    for nm, (dst, src) in (("h2d", (dv, hp)), ("d2h", (hp, dv))):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        out[nm + "_GBps"] = n / best / 1e9
    pg = np.ones(n, np.uint8)
    t = time.perf_counter(); dv.copy_(torch.from_numpy(pg)); torch.cuda.synchronize()
    out["pageable_h2d_GBps"] = n / (time.perf_counter() - t) / 1e9
    # what the host can stream into a pinned buffer with one thread (the pipeline's worker threads each do this)
    src = np.ones(n // 8, np.float64)
    dstp = hp.numpy()[: n // 2].view(np.float32)
    t = time.perf_counter(); np.copyto(dstp, src, casting="same_kind"); dt1 = time.perf_counter() - t
    out["numpy_f64_to_f32_into_pinned_one_thread_GBps_of_source"] = (n) / dt1 / 1e9
    emit(config="bare transfers, 512 MiB", cpus=os.cpu_count(), **out)
    del hp, dv

if "a11" in which:
    one_shot("a11 sparse_nmf 513x72000 r=100 KL full, 100 it (run_basis_train.m:88)", 513, 72_000, 100, 100, np.float64)
if "c2" in which:
    one_shot("C2 sparse_nmf 257x100000 r=256 KL full, 200 it", 257, 100_000, 256, 200, np.float64)
if "mel" in which:
    one_shot("mel sparse_nmf 64x72000 r=100 KL full, 100 it (run_basis_train.m:91)", 64, 72_000, 100, 100, np.float64)

if "c4" in which:
    F, T, Rx, Rd = 513, 100_000, 100, 100
    X, _, _ = synth(F, T, Rx, 1); D, _, _ = synth(F, T, Rd, 2)
    Y = np.asfortranarray(X + D + 1e-9)
    B = np.asfortranarray(np.random.default_rng(3).random((F, Rx + Rd)))
    p = dict(cf="kl", sparsity=5, max_iter=50, conv_eps=0, cost_check=1, random_seed=1)
    run_basis_dnmf(Y[:, :4096], X[:, :4096], D[:, :4096], B, Rx, Rd, p, ctx=ctx)  # warm: code objects, pinned buffers, second stream
    # rand(r, n) of solve 1 (src/sparse_nmf.m:133-134; 20 M uniforms) is drawn by the CALLER (MATLAB's generator in the wrapper,
    # NumPy's here), outside the timed region, and handed over as an array: 160 MB more to upload.  "device rng" rows let the
    # engine draw it in HBM instead.
    t = time.perf_counter(); H0h = np.asfortranarray(np.random.RandomState(1).random_sample((Rx + Rd, T))); rng_s = time.perf_counter() - t
    res = {}
    for dtype in (np.float64, np.float32):
        Yd, Xd, Dd, H0d = (np.asfortranarray(M, dtype=dtype) for M in (Y, X, D, H0h))
        for mode, kw in (("three sparse_nmf calls", dict(resident=False, h0=H0d)), ("one resident call", dict(resident=True, h0=H0d)),
                         ("one resident call, device rng", dict(resident=True, h0="device"))):
            best = None
            for _ in range(2):
                ctx.xfer_stats(reset=True)
                t = time.perf_counter()
                Bh, Ah = run_basis_dnmf(Yd, Xd, Dd, B, Rx, Rd, p, ctx=ctx, dtype=dtype, **kw)
                dt = time.perf_counter() - t
                st = ctx.xfer_stats()
                if best is None or dt < best[0]:
                    best = (dt, st)
            dt, st = best
            res[(np.dtype(dtype).name, mode)] = Bh
            emit(config=f"C4 run_basis_DNMF 3 solves x 50 it, {F}x{T}, R_x=R_d=100: {mode}, A_hat returned", host_dtype=np.dtype(dtype).name,
                 call_s=dt, solver_iterations_per_s=150 / dt, h2d_s_primary_ctx=st["h2d_wall_s"], h2d_MB_primary_ctx=st["h2d_bytes"] / 1e6,
                 d2h_s=st["d2h_wall_s"], d2h_MB=st["d2h_bytes"] / 1e6, host_rand_draw_outside_the_call_s=rng_s,
                 note="h2d of X and D runs on the second context of the resident path (under solve 1) and is not in the primary counters")
        assert np.array_equal(res[(np.dtype(dtype).name, "three sparse_nmf calls")], res[(np.dtype(dtype).name, "one resident call")])
    # B_hat only (what run_basis_DNMF.m returns): A_hat never leaves HBM
    from se_snmf_nat_amd.api import _run_basis_dnmf_resident
    for dtype in (np.float64, np.float32):
        Yd, Xd, Dd = (np.asfortranarray(M, dtype=dtype) for M in (Y, X, D))
        best = 1e9
        for _ in range(2):
            t = time.perf_counter()
            _run_basis_dnmf_resident(Yd, Xd, Dd, B, Rx, Rd, p, ctx=ctx, dtype=dtype, h0="device", want_a=False)
            best = min(best, time.perf_counter() - t)
        emit(config=f"C4 run_basis_DNMF, one resident call, B_hat only (the reference's return value)", host_dtype=np.dtype(dtype).name,
             call_s=best, solver_iterations_per_s=150 / best)
    # the three solves with everything resident: the floor of any call
    solve = 0.0
    _, _, H0 = synth(F, T, Rx + Rd, 5, np.float32)
    solve += resident_solve_s(F, T, Rx + Rd, 50, Y, B, H0, beta=1.0, sparsity=5.0, w_update_ind=np.zeros(Rx + Rd, bool))
    for M, r0 in ((X, 0), (D, Rx)):
        solve += resident_solve_s(F, T, Rx, 50, M, B[:, r0:r0 + Rx], H0[r0:r0 + Rx], beta=1.0, sparsity=5.0, h_update_ind=np.zeros(Rx, bool))
    emit(config="C4 the three solves alone, data resident", resident_solve_s=solve, solver_iterations_per_s=150 / solve)
    # from the waveforms: audio in, B_hat out (run_basis_DNMF.m:1 as the reference calls it)
    from se_snmf_nat_amd import frontend as fe, train
    fp = dict(fe.default_params(), cf="kl", sparsity=5, max_iter=50, conv_eps=0, cost_check=1, random_seed=1, R_x=Rx, R_d=Rd)
    n = 160 * T + 1024 + 1
    rs = np.random.RandomState(0)
    xs, ds = (rs.randn(n) * 3000).astype(np.float32), (rs.randn(n) * 1000).astype(np.float32)
    assert fe.num_frames(n, fp) == T
    train.run_basis_DNMF(xs[:200_000], ds[:200_000], B, fp, ctx=ctx, h0="device")
    best = 1e9
    for _ in range(2):
        t = time.perf_counter(); Bh = train.run_basis_DNMF(xs, ds, B, fp, ctx=ctx, h0="device"); best = min(best, time.perf_counter() - t)
    emit(config=f"C4 run_basis_DNMF(x, d, B, p) from the waveforms ({n} samples -> {F}x{T}), B_hat out", call_s=best,
         solver_iterations_per_s=150 / best, audio_MB=2 * n * 4 / 1e6)

if "c4mel" in which:
    # run_basis_DNMF_Mel.m:75-88 -- the same three solves on 64 Mel bands (H-only at r = 200: k_hstep_sf, W-only x 2 at r = 100: k_wstats_sf)
    from se_snmf_nat_amd.api import _run_basis_dnmf_resident
    F, T, Rx, Rd = 64, 100_000, 100, 100
    X, _, _ = synth(F, T, Rx, 1); D, _, _ = synth(F, T, Rd, 2)
    Y = np.asfortranarray(X + D + 1e-9)
    B = np.asfortranarray(np.random.default_rng(3).random((F, Rx + Rd)))
    p = dict(cf="kl", sparsity=5, max_iter=50, conv_eps=0, cost_check=1, random_seed=1)
    run_basis_dnmf(Y[:, :4096], X[:, :4096], D[:, :4096], B, Rx, Rd, p, ctx=ctx)
    for dtype in (np.float64, np.float32):
        Yd, Xd, Dd = (np.asfortranarray(M, dtype=dtype) for M in (Y, X, D))
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            _run_basis_dnmf_resident(Yd, Xd, Dd, B, Rx, Rd, p, ctx=ctx, dtype=dtype, h0="device", want_a=False)
            best = min(best, time.perf_counter() - t)
        emit(config=f"C4-Mel run_basis_DNMF_Mel 3 solves x 50 it, {F}x{T}, R_x=R_d=100: one resident call, B_hat only", host_dtype=np.dtype(dtype).name,
             call_s=best, solver_iterations_per_s=150 / best)
    solve = 0.0
    _, _, H0 = synth(F, T, Rx + Rd, 5, np.float32)
    solve += resident_solve_s(F, T, Rx + Rd, 50, Y, B, H0, beta=1.0, sparsity=5.0, w_update_ind=np.zeros(Rx + Rd, bool))
    for M, r0 in ((X, 0), (D, Rx)):
        solve += resident_solve_s(F, T, Rx, 50, M, B[:, r0:r0 + Rx], H0[r0:r0 + Rx], beta=1.0, sparsity=5.0, h_update_ind=np.zeros(Rx, bool))
    emit(config="C4-Mel the three solves alone, data resident", resident_solve_s=solve, solver_iterations_per_s=150 / solve)

if "c4m" in which:
    # The device-list entry of run_basis_DNMF (round 5): every rank's shard through its own pinned pipeline, A_hat resident per rank.
    # With n ranks on ONE device the shards' kernels serialise, so the yardstick is n x (the three solves of one shard, data resident).
    from se_snmf_nat_amd.api import _run_basis_dnmf_resident
    F, T, Rx, Rd = 513, 100_000, 100, 100
    X, _, _ = synth(F, T, Rx, 1); D, _, _ = synth(F, T, Rd, 2)
    Y = np.asfortranarray(X + D + 1e-9)
    B = np.asfortranarray(np.random.default_rng(3).random((F, Rx + Rd)))
    p = dict(cf="kl", sparsity=5, max_iter=50, conv_eps=0, cost_check=1, random_seed=1)
    _, _, H0 = synth(F, T, Rx + Rd, 5, np.float32)
    for n in (2, 4, 8):
        devs = [0] * n
        Ts = T // n
        shard = resident_solve_s(F, Ts, Rx + Rd, 50, Y[:, :Ts], B, H0[:, :Ts], beta=1.0, sparsity=5.0, w_update_ind=np.zeros(Rx + Rd, bool))
        for M, r0 in ((X, 0), (D, Rx)):
            shard += resident_solve_s(F, Ts, Rx, 50, M[:, :Ts], B[:, r0:r0 + Rx], H0[r0:r0 + Rx, :Ts], beta=1.0, sparsity=5.0, h_update_ind=np.zeros(Rx, bool))
        _run_basis_dnmf_resident(Y[:, :8192], X[:, :8192], D[:, :8192], B, Rx, Rd, p, ctx=None, dtype=np.float64, h0="device", want_a=False, devices=devs)  # the team exists
        best = {}
        for want_a in (False, True):
            b = 1e9
            for _ in range(2):
                t = time.perf_counter()
                _run_basis_dnmf_resident(Y, X, D, B, Rx, Rd, p, ctx=None, dtype=np.float64, h0="device", want_a=want_a, devices=devs)
                b = min(b, time.perf_counter() - t)
            best[want_a] = b
        # round 4's form of the same call: three snmf_sparse_nmf_multi_* calls, A_hat through the host in between
        t = time.perf_counter(); run_basis_dnmf(Y, X, D, B, Rx, Rd, p, devices=devs, resident=False, h0="device"); three = time.perf_counter() - t
        emit(config=f"C4 run_basis_DNMF over a device list: {n} ranks on device 0, {F}x{T} sharded, 3 solves x 50 it, fp64 host arrays",
             ranks=n, call_s_B_hat_only=best[False], call_s_with_A_hat=best[True], three_multi_calls_s=three,
             n_times_one_shard_resident_s=n * shard, one_shard_resident_s=shard, call_over_n_shards=best[False] / (n * shard),
             solver_iterations_per_s=150 / best[False])
