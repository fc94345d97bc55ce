"""Dev measurement: what ONE rank of the frame-sharded bench costs per iteration at the shard sizes of 1, 2, 4 and 8
GPUs (257 x 100000/N, r = 256, KL), on a single GPU.  The collective is a one-rank RCCL group (an identity sum, so the
call path, stream ordering and host overhead of bench.py --gpus N are all there; the xGMI transfer itself is not).
Prints per-N: ms/iteration through (a) the C loop without any exchange, (b) the sharded loop with the RCCL call, and
the strong-scaling bound each implies against N = 1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    from bench import make_problem, SPARSITY
    from se_snmf_nat_amd import Context, Plan
    from se_snmf_nat_amd.dist import ShardedTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    F, T, r, K, W = 257, 100_000, 256, 200, 300  # W: enough load (>= 30 ms) for the chip to reach its steady clock at every shard size
    base = {}
    shard_ms = {}
    for n in (1, 2, 4, 8):
        Tn = T // n
        V, W0, H0 = make_problem(F, T, r, 0, Tn)
        ctx = Context(0)
        plan = Plan(ctx, F, Tn, r, beta=1.0, max_iter=W + K + 21, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
        plan.set_v(V.astype(np.float32)); plan.set_w(W0); plan.set_h(H0.astype(np.float32)); plan.init()
        plan.run_async(W); ctx.sync()
        t = time.perf_counter(); plan.run_async(K); ctx.sync(); a = (time.perf_counter() - t) / K * 1e3
        ctx.timing(True); plan.run_async(20); ctx.sync()
        fam = {f: round(ctx.timing_get(f)[0] * 1e3, 1) for f in ("hstep", "wstats", "reduce", "wapply", "wfin")}
        ctx.timing(False)
        geo = plan.describe()
        plan.close()
        tr = ShardedTrainer(V.astype(np.float32), W0, H0.astype(np.float32), beta=1.0, sparsity=SPARSITY, max_iter=W + K + 1,
                            conv_eps=0.0, cost_check=True, device=0)
        tr.world = 2  # take the RCCL branch (one-rank group: identity)
        tr.run(W); tr.sync()
        t = time.perf_counter(); tr.run(K); th = time.perf_counter() - t; tr.sync(); b = (time.perf_counter() - t) / K * 1e3
        base.setdefault("a", a); base.setdefault("b", b)
        shard_ms[n] = a
        print(f"N={n}: shard {F}x{Tn}  C loop {a:.4f} ms/it (bound {base['a'] / a:.2f}x)   sharded loop + RCCL call {b:.4f} ms/it "
              f"(bound {base['b'] / b:.2f}x; host issue {th / K * 1e3:.4f} ms/it)  kernels us {fam}", flush=True)
        if n == 8:
            print("   ", geo)
    dist.destroy_process_group()
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


if __name__ == "__main__":
    main()
