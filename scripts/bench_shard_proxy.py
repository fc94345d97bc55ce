"""Dev measurement: what ONE rank of a frame-sharded solve costs per iteration at the shard sizes of 1, 2, 4 and 8 GPUs, on a
single GPU, for the configs whose strong scaling SURVEY.md section 8d asks about:
  c2   257 x 100000 / N, r = 256, KL, full update                 (BASELINE configs[1], the headline)
  c5   513 x 500000 / N, r = 512, beta = 2, lambda = 50           (BASELINE configs[4])
  c4h  513 x 800000 / N, r = 200, KL, H-only                      (run_basis_DNMF.m:40 on 8 x 100000 frames: configs[3])
  c4w  513 x 800000 / N, r = 100, KL, W-only                      (run_basis_DNMF.m:47,53)
The collective is a one-rank RCCL group (an identity sum, so the call path, stream ordering and host overhead of
bench.py --gpus N are all there; the xGMI transfer itself is not).  Per N: ms/iteration through (a) the C loop without any
exchange (snmf_plan_run: k_wfin fuses the chunk reduction and the W update), (b) the sharded step-API loop with the RCCL
call; the strong-scaling bound each implies against N = 1; and a WHAT-IF column = (a) + the fixed part the one-process
multi-device entry adds per rank and iteration (measured at the bottom with n ranks on this one device, EVENTS ordering).
    python scripts/bench_shard_proxy.py [c2 c5 c4h c4w]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CONFIGS = {
    "c2": dict(F=257, T=100_000, r=256, beta=1.0, sparsity=5.0, mode="full", K=200, W=300),
    "c5": dict(F=513, T=500_000, r=512, beta=2.0, sparsity=50.0, mode="full", K=10, W=6),
    "c4h": dict(F=513, T=800_000, r=200, beta=1.0, sparsity=5.0, mode="h", K=40, W=40),
    "c4w": dict(F=513, T=800_000, r=100, beta=1.0, sparsity=5.0, mode="w", K=60, W=60),
}


def synth(F, T, r, seed=0):
    rd = np.random.default_rng(seed)
    Wt = rd.gamma(0.5, 1.0, size=(F, 16)).astype(np.float32)
    V = np.empty((F, T), np.float32, order="F")
    H0 = np.empty((r, T), np.float32, order="F")
    for t0 in range(0, T, 50000):
        t1 = min(T, t0 + 50000)
        V[:, t0:t1] = Wt @ rd.gamma(0.3, 1.0, size=(16, t1 - t0)).astype(np.float32) + 1e-9
        H0[:, t0:t1] = rd.random((r, t1 - t0), dtype=np.float32)
    return V, rd.random((F, r)), H0


def proxy(name, c, fixed_us):
    """rows for one config; fixed_us[n]: what the multi-device entry adds per rank and iteration (None: not measured yet)"""
    from se_snmf_nat_amd import Context, Plan
    from se_snmf_nat_amd.dist import ShardedTrainer
    F, T, r, K, W = c["F"], c["T"], c["r"], c["K"], c["W"]
    kw = {}
    if c["mode"] == "h":
        kw["w_update_ind"] = np.zeros(r, bool)
    if c["mode"] == "w":
        kw["h_update_ind"] = np.zeros(r, bool)
    base = {}
    out = {}
    for n in (1, 2, 4, 8):
        Tn = T // n
        V, W0, H0 = synth(F, Tn, r)
        ctx = Context(0)
        plan = Plan(ctx, F, Tn, r, beta=c["beta"], max_iter=W + K + 21, conv_eps=0.0, cost_check=True, sparsity=c["sparsity"], **kw)
        plan.set_v(V); plan.set_w(W0); plan.set_h(H0); plan.init()
        plan.run_async(W); ctx.sync()
        t = time.perf_counter(); plan.run_async(K); ctx.sync(); a = (time.perf_counter() - t) / K * 1e3
        ctx.timing(True); plan.run_async(min(20, K)); ctx.sync()
        fam = {f: round(ctx.timing_get(f)[0] * 1e3, 1) for f in ("hstep", "wstats", "reduce", "wapply", "wfin") if ctx.timing_get(f)[1]}
        ctx.timing(False)
        plan.close()
        ctx.close()
        tr = ShardedTrainer(V, W0, H0, beta=c["beta"], sparsity=c["sparsity"], max_iter=W + 2 * K + 41, conv_eps=0.0, cost_check=True, device=0,
                            w_update_ind=kw.get("w_update_ind"), h_update_ind=kw.get("h_update_ind"))
        tr.world = 2  # take the RCCL branch (one-rank group: identity)
        tr.run(W); tr.sync()
        t = time.perf_counter(); tr.run(K); th = time.perf_counter() - t; tr.sync(); b = (time.perf_counter() - t) / K * 1e3
        # (round 6) ... and with the collective issued by the library itself: ncclAllReduce from C on the engine's stream (snmf_plan_run_sharded_rccl)
        nat = ""
        if tr.use_native_rccl(force_single=True):
            tr.run(20); tr.sync()
            t = time.perf_counter(); tr.run(K); thn = time.perf_counter() - t; tr.sync(); bn = (time.perf_counter() - t) / K * 1e3
            nat = f"   step loop + RCCL call FROM C {bn:.4f} ms/it ({(bn - a) * 1e3:+.1f} us against the C loop; host issue {thn / K * 1e3:.4f} ms/it)"
        del tr
        base.setdefault("a", a); base.setdefault("b", b)
        out[n] = a
        wi = ""
        if fixed_us.get(n) is not None and c["mode"] != "h":
            w_ms = a + fixed_us[n] * 1e-3
            wi = f"  what-if (C loop + the multi entry's fixed part {fixed_us[n]:+.1f} us): {w_ms:.4f} ms/it = {base['a'] / w_ms:.2f}x"
        print(f"{name} N={n}: shard {F}x{Tn} r={r}  C loop {a:.4f} ms/it (bound {base['a'] / a:.2f}x)   step-API loop + RCCL call {b:.4f} ms/it "
              f"(bound {base['b'] / b:.2f}x; host issue {th / K * 1e3:.4f} ms/it){nat}  kernels us {fam}{wi}", flush=True)
    return out


def main():
    import torch
    import torch.distributed as dist
    from bench import make_problem, SPARSITY
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    which = [a for a in sys.argv[1:] if a in CONFIGS] or ["c2", "c5", "c4h", "c4w"]
    F, T, r, K, W = 257, 100_000, 256, 200, 300
    fixed_us = {}
    if "c2" in which:
        fixed_us = multi_fixed_part(make_problem, SPARSITY, F, T, r, K, W)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    for name in which:
        proxy(name, CONFIGS[name], fixed_us)
    dist.destroy_process_group()


def multi_fixed_part(make_problem, SPARSITY, F, T, r, K, W):
    """what the one-process multi-device entry adds per rank and iteration on C2 (us), by number of ranks"""
    from se_snmf_nat_amd import Context, Plan
    shard_ms = {}
    for n in (2, 4, 8):
        V, W0, H0 = make_problem(F, T, r, 0, T // n)
        ctx = Context(0)
        plan = Plan(ctx, F, T // n, r, beta=1.0, max_iter=W + K + 1, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
        plan.set_v(V.astype(np.float32)); plan.set_w(W0); plan.set_h(H0.astype(np.float32)); plan.init()
        plan.run_async(W); ctx.sync()
        t = time.perf_counter(); plan.run_async(K); ctx.sync(); shard_ms[n] = (time.perf_counter() - t) / K * 1e3
        plan.close()
        ctx.close()
    fixed = {1: 0.0}
    # ---- the one-process multi-GPU entry (snmf_multi_*) with n ranks on THIS one device: the ranks' kernels serialise on
    # the GPU, so ms/iteration minus n x (C loop of one shard) is what the entry adds per iteration: exchange kernels,
    # ordering, host issue.  EVENTS = hipEvents + host barrier per iteration (what ranks that share a device get);
    # FLAGS = device-side arrival words, the host only enqueues (what ranks with a device of their own get; on one device
    # it is only safe while every rank's stream has a hardware queue of its own, so n = 2 here).
    import ctypes as C
    from se_snmf_nat_amd import _lib
    from se_snmf_nat_amd.api import _make_params
    lib = _lib.load()
    V, W0, H0 = make_problem(F, T, r)
    Vf, Hf, Wf = np.asfortranarray(V, np.float32), np.asfortranarray(H0, np.float32), np.asfortranarray(W0)
    for n, mode in ((2, "events"), (2, "flags"), (4, "events"), (8, "events")):
        sp = _make_params(F, T, r, 1.0, W + K + 1, 0.0, 1, True, 0, SPARSITY, None, None)
        h = C.c_void_p()
        devs = np.zeros(n, np.int32)
        _lib.check(lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), n, C.byref(sp), None, C.byref(h)))
        _lib.check(lib.snmf_multi_set_exchange(h, _lib.EXCHANGE_FLAGS if mode == "flags" else _lib.EXCHANGE_EVENTS))
        _lib.check(lib.snmf_multi_set_v_f32(h, C.c_void_p(Vf.ctypes.data), F))
        _lib.check(lib.snmf_multi_set_w_f64(h, C.c_void_p(Wf.ctypes.data), F))
        _lib.check(lib.snmf_multi_set_h_f32(h, C.c_void_p(Hf.ctypes.data), r))
        _lib.check(lib.snmf_multi_init(h))
        done = C.c_int32()
        _lib.check(lib.snmf_multi_run(h, W, C.byref(done)))
        t = time.perf_counter(); _lib.check(lib.snmf_multi_run(h, K, C.byref(done))); ms = (time.perf_counter() - t) / K * 1e3
        lib.snmf_multi_destroy(h)
        print(f"multi C-ABI, {n} ranks on one device, {mode}: {ms:.4f} ms/it for the whole problem = {n} x {ms / n:.4f}; "
              f"one shard's C loop {shard_ms[n]:.4f} -> the entry adds {(ms / n - shard_ms[n]) * 1e3:.1f} us per rank and iteration", flush=True)
        if mode == "events":
            fixed[n] = (ms / n - shard_ms[n]) * 1e3
    return fixed


if __name__ == "__main__":
    main()
