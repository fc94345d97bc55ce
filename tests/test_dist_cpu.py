"""world_size-2 gloo test of the frame-sharded loop (se_snmf_nat_amd/dist.py) on CPU.

The product engine is the HIP plan; here the same ShardedLoop drives a NumPy engine
(tests/oracle_engine.py) so that the sharding algebra -- one sum-all-reduce of
[G | s | div | sum(S.*H)] per iteration, identical W epilogue on every rank, delayed objective,
stop-index recovery -- is checked against the unsharded oracle without a GPU.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    import torch
    import torch.distributed as dist
    from oracle.sparse_nmf_oracle import synth_problem
    from oracle_engine import OracleEngine
    from se_snmf_nat_amd.dist import ShardedLoop, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F, T, r = case["F"], case["T"], case["r"]
    V, W0, H0 = synth_problem(F, T, r)
    t0, t1 = shard_bounds(T, world, rank)
    eng = OracleEngine(V[:, t0:t1], W0, H0[:, t0:t1], beta=case["beta"], sparsity=case["sparsity"],
                       max_iter=case["max_iter"], conv_eps=case["conv_eps"], cost_check=True,
                       w_ind=case.get("w_ind"), h_ind=case.get("h_ind"))
    stats_t = torch.zeros(eng.stats_len(), dtype=torch.float64)
    stats = stats_t.numpy()

    class Eng:  # adapt pointer-taking step API to the numpy view
        def __getattr__(self, k):
            return getattr(eng, k)

        def wstats(self, _):
            eng.wstats(stats)

        def wapply(self, _):
            eng.wapply(stats)

        def objstats(self, _):
            eng.objstats(stats)

        def objapply(self, _):
            eng.objapply(stats)

    loop = ShardedLoop(Eng(), stats_t, lambda t: dist.all_reduce(t), max_iter=case["max_iter"],
                       can_stop=case["conv_eps"] > 0, cost_check=True, poll_every=3)
    loop.run()
    q.put((rank, eng.w, eng.h, np.array(eng.cost_hist), eng.n_iter, (t0, t1)))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    dict(F=33, T=101, r=7, beta=1.0, sparsity=0.7, max_iter=12, conv_eps=0.0),
    dict(F=33, T=101, r=7, beta=1.0, sparsity=0.7, max_iter=60, conv_eps=2e-3),
    dict(F=20, T=64, r=5, beta=2.0, sparsity=0.3, max_iter=10, conv_eps=0.0),
    dict(F=20, T=64, r=5, beta=0.0, sparsity=0.01, max_iter=10, conv_eps=0.0),
    dict(F=20, T=64, r=6, beta=1.0, sparsity=0.3, max_iter=10, conv_eps=0.0, w_ind=[0, 0, 0, 1, 1, 1]),
    dict(F=20, T=64, r=6, beta=1.0, sparsity=0.3, max_iter=30, conv_eps=1e-3, w_ind=[0] * 6),       # H-only
    dict(F=20, T=64, r=6, beta=1.0, sparsity=0.3, max_iter=30, conv_eps=1e-3, h_ind=[0] * 6),       # W-only
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"b{c['beta']}-eps{c['conv_eps']}-{'w' if 'w_ind' in c else ''}{'h' if 'h_ind' in c else ''}")
def test_sharded_loop_matches_unsharded_oracle(case):
    import torch.multiprocessing as mp
    from oracle.sparse_nmf_oracle import sparse_nmf, synth_problem
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, case, q)) for rk in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    V, W0, H0 = synth_problem(case["F"], case["T"], case["r"])
    cf = {1.0: "kl", 2.0: "ed", 0.0: "is"}[case["beta"]]
    p = dict(cf=cf, sparsity=case["sparsity"], max_iter=case["max_iter"], conv_eps=case["conv_eps"], init_w=W0,
             init_h=H0, cost_check=1)
    if "w_ind" in case:
        p["w_update_ind"] = np.array(case["w_ind"], bool)
    if "h_ind" in case:
        p["h_update_ind"] = np.array(case["h_ind"], bool)
    w, h, o = sparse_nmf(V, p)
    H = np.concatenate([r_[2] for r_ in res], axis=1)
    for r_ in res:
        assert r_[4] == o["n_iter"]
        np.testing.assert_allclose(r_[1], w, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r_[3], o["cost"], rtol=1e-10)
    # replicas of W bit-identical across ranks
    assert np.array_equal(res[0][1], res[1][1])
    np.testing.assert_allclose(H, h, rtol=1e-9, atol=1e-12)


def test_shard_bounds_cover():
    from se_snmf_nat_amd.dist import shard_bounds
    for T in (1, 7, 100000):
        for ws in (1, 2, 3, 8):
            b = [shard_bounds(T, ws, r) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == T
            assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))


def test_sharded_initial_activations_are_the_unsharded_draw():
    """dist.h0_columns: the column blocks the ranks draw for rand(r, n) of solve 1 (src/sparse_nmf.m:133-134) are the columns of
    the ONE matrix the unsharded mirror draws, for any world size -- so the sharded DNMF loop does not depend on the number of GPUs."""
    from se_snmf_nat_amd.dist import h0_columns, shard_bounds
    r, T, seed = 7, 101, 3
    full = np.random.RandomState(seed).random_sample((r, T))
    for world in (1, 2, 3, 8):
        parts = []
        for rank in range(world):
            t0, t1 = shard_bounds(T, world, rank)
            parts.append(h0_columns(seed, r, t0, t1 - t0, T))
        assert np.array_equal(np.concatenate(parts, axis=1), full)
