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
    for n in (1, 2, 4, 8):
        Tn = T // n
        V, W0, H0 = make_problem(F, T, r, 0, Tn)
        ctx = Context(0)
        plan = Plan(ctx, F, Tn, r, beta=1.0, max_iter=W + K + 21, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
        plan.set_v(V.astype(np.float32)); plan.set_w(W0); plan.set_h(H0.astype(np.float32)); plan.init()
        plan.run_async(W); ctx.sync()
        t = time.perf_counter(); plan.run_async(K); ctx.sync(); a = (time.perf_counter() - t) / K * 1e3
        ctx.timing(True); plan.run_async(20); ctx.sync()
        fam = {f: round(ctx.timing_get(f)[0] * 1e3, 1) for f in ("hstep", "wstats", "reduce", "wapply")}
        ctx.timing(False)
        geo = plan.describe()
        plan.close()
        tr = ShardedTrainer(V.astype(np.float32), W0, H0.astype(np.float32), beta=1.0, sparsity=SPARSITY, max_iter=W + K + 1,
                            conv_eps=0.0, cost_check=True, device=0)
        tr.world = 2  # take the RCCL branch (one-rank group: identity)
        tr.run(W); tr.sync()
        t = time.perf_counter(); tr.run(K); th = time.perf_counter() - t; tr.sync(); b = (time.perf_counter() - t) / K * 1e3
        base.setdefault("a", a); base.setdefault("b", b)
        print(f"N={n}: shard {F}x{Tn}  C loop {a:.4f} ms/it (bound {base['a'] / a:.2f}x)   sharded loop + RCCL call {b:.4f} ms/it "
              f"(bound {base['b'] / b:.2f}x; host issue {th / K * 1e3:.4f} ms/it)  kernels us {fam}", flush=True)
        if n == 8:
            print("   ", geo)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
