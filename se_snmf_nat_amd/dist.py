"""Frame-sharded multi-GPU basis training.

The reference is single-process MATLAB (run_basis_train.m:80-91, run_basis_DNMF.m:36-55 call
sparse_nmf on one concatenated spectrogram).  Columns (frames) of V and H are independent given
W, so the frame axis is partitioned into contiguous blocks, one process per GPU, W replicated.
Per iteration there is exactly ONE exchange: a sum-all-reduce (RCCL over xGMI through
torch.distributed) of the fp64 statistics buffer

    [ (V./Lam)H' or Q (F*r) | P (F*r, beta != 1) | rowsum(H) (r) | div | sum(S.*H) ]

after which every rank applies the identical deterministic W epilogue + convergence test
(src/sparse_nmf.m:215-244, :272-284), so the replicas of W stay bit-identical.  H-only solves
need no exchange except the two cost scalars.

The loop below is engine-agnostic: `engine` is anything with the step interface of
`se_snmf_nat_amd.api.Plan` (hstep / wstats / wapply / objstats / objapply / stopped).  The product
engine is the HIP plan; tests/ drive the same loop over gloo with an oracle-backed engine to pin
the sharding algebra on CPU.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(T, world_size, rank):
    """Contiguous, balanced frame range [t0, t1) of `rank`."""
    t0 = (T * rank) // world_size
    t1 = (T * (rank + 1)) // world_size
    return t0, t1


class ShardedLoop:
    """for it = 1:max_iter of src/sparse_nmf.m:186 with the frame axis sharded over ranks."""

    def __init__(self, engine, stats, all_reduce, *, max_iter, can_stop, cost_check, poll_every=4, step_view=None):
        """engine: step interface; stats: buffer object the engine fills (torch tensor or numpy);
        all_reduce(view): in-place sum over ranks of the given slice of `stats` (no-op for world_size 1);
        step_view: the slice one iteration exchanges (default: all of it; H-only solves: the two cost scalars at the tail)."""
        self.e = engine
        self.stats = stats
        self.all_reduce = all_reduce
        self.step_view = stats if step_view is None else step_view
        self.max_iter = int(max_iter)
        self.can_stop = bool(can_stop)
        self.cost_check = bool(cost_check)
        self.poll_every = int(poll_every)
        self.it = 0
        self.finalized = False
        self.rccl_comm = None  # an api.RcclComm: the product engine's loop then calls RCCL from C (ShardedTrainer sets it)

    def _ptr(self):
        s = self.stats
        return s.data_ptr() if hasattr(s, "data_ptr") else s.ctypes.data

    def step(self):
        self.e.hstep()
        self.e.wstats(self._ptr())
        self.all_reduce(self.step_view)
        self.e.wapply(self._ptr())
        self.it += 1

    def run(self, n_iters=None):
        target = self.max_iter if n_iters is None else min(self.max_iter, self.it + int(n_iters))
        if hasattr(self.e, "run_sharded") and hasattr(self.stats, "data_ptr"):
            # the product engine: the loop runs inside the library (snmf_plan_run_sharded), one call for all the iterations;
            # the collective comes back as a callback on the (whole or two-scalar) statistics buffer
            base, esz = self.stats.data_ptr(), self.stats.element_size()

            def ar(ptr, n):  # exactly the `n` doubles at `ptr` the library asks for (include/snmf.h: snmf_allreduce_fn)
                off = (int(ptr) - base) // esz
                if off < 0 or off + int(n) > self.stats.numel() or (int(ptr) - base) % esz:
                    raise ValueError("collective asked for a range outside the statistics buffer")
                self.all_reduce(self.stats[off:off + int(n)])
            last = target >= self.max_iter
            fin = last and self.cost_check and not self.finalized
            if self.rccl_comm is not None:
                # the collective issued by the library itself (ncclAllReduce on the engine's stream, csrc/snmf_tu_rccl.hip): no callback
                # into Python per iteration (29 us of host work: a quarter of an iteration on a 12 500-frame shard of BASELINE configs[1])
                ran = self.e.run_sharded_rccl(target - self.it, self._ptr(), self.rccl_comm, poll_every=self.poll_every if self.can_stop else 0,
                                              finalize=fin)
            else:
                ran = self.e.run_sharded(target - self.it, self._ptr(), ar, poll_every=self.poll_every if self.can_stop else 0, finalize=fin)
            self.it += ran
            if last and self.it >= self.max_iter and self.cost_check and self.it > 0:
                self.finalized = True
            return self.it
        since = 0
        stopped = False
        while self.it < target:
            self.step()
            since += 1
            if self.can_stop and since >= self.poll_every:
                since = 0
                if self.e.stopped():
                    stopped = True
                    break
        if not stopped and self.it >= self.max_iter and self.cost_check and not self.finalized and self.it > 0:
            self.e.objstats(self._ptr())
            self.all_reduce(self.stats[-2:])  # (div, sum S.*H): all the final objective needs
            self.e.objapply(self._ptr())
            self.finalized = True
        return self.it


class ShardedTrainer:
    """One rank of a frame-sharded solve on its own MI355X (one process per GPU).

    v_local: F x T_local block of the spectrogram (numpy, or a torch CUDA tensor shaped
    (T_local, F) i.e. column-major), w0: F x r (replicated), h0_local: r x T_local.
    `group`: torch.distributed process group (None = default).  world_size 1 needs no init.
    """

    def __init__(self, v_local, w0, h0_local, *, beta=1.0, sparsity=0.0, max_iter=100, conv_eps=0.0,
                 cost_check=True, floor_v=True, w_update_ind=None, h_update_ind=None, device=0, group=None,
                 use_torch_stream=True):
        import torch
        import torch.distributed as dist
        from .api import Context, Plan
        self.torch = torch
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = Context(device)
        # The engine's kernels and the collective must be ordered on ONE stream.  torch's default
        # stream is handle 0, which snmf_ctx_set_stream reads as "own stream", so the trainer owns
        # a side stream, hands it to the engine and issues every all-reduce under it (RCCL then
        # orders its own stream against it with events, as it does for any torch op).
        self.stream = torch.cuda.Stream(self.device) if use_torch_stream else None
        if self.stream is not None:
            self.ctx.set_stream(self.stream.cuda_stream)
        F = w0.shape[0] if not hasattr(w0, "data_ptr") else w0.shape[1]
        r = w0.shape[1] if not hasattr(w0, "data_ptr") else w0.shape[0]
        T = v_local.shape[1] if not hasattr(v_local, "data_ptr") else v_local.shape[0]
        self.plan = Plan(self.ctx, F, T, r, beta=beta, max_iter=max_iter, conv_eps=conv_eps, cost_check=cost_check,
                         floor_v=floor_v, sparsity=sparsity, w_update_ind=w_update_ind, h_update_ind=h_update_ind)
        self.plan.set_v(v_local)
        self.plan.set_w(w0)
        self.plan.set_h(h0_local)
        self.plan.init()
        self.stats = torch.zeros(self.plan.stats_len(), dtype=torch.float64, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()  # the zero fill ran on torch's current stream
        # H-only solves exchange nothing but the two cost scalars at the tail of the buffer
        w_any = True if w_update_ind is None else bool(np.asarray(w_update_ind).any())
        self.loop = ShardedLoop(self.plan, self.stats, self._all_reduce, max_iter=max_iter,
                                can_stop=bool(cost_check) and conv_eps > 0, cost_check=cost_check,
                                step_view=self.stats if w_any else self.stats[-2:])
        self.rccl = None
        if self.world > 1 and self.stream is not None:
            self.use_native_rccl()

    def use_native_rccl(self, force_single=False):
        """Route the per-iteration sum through the library's own ncclAllReduce (api.RcclComm) instead of the torch.distributed
        callback: RCCL backend only (a gloo dry run keeps the callback), SNMF_RCCL_NATIVE=0 opts out.  The communicator's id travels
        from rank 0 over the torch.distributed group the trainer was given.  force_single: a one-rank communicator (tests: the
        call path on a one-GPU box, where RCCL refuses two ranks on one device).  Returns True when the native path is in use."""
        import os
        from .api import RcclComm
        if os.environ.get("SNMF_RCCL_NATIVE", "1") == "0" or not RcclComm.available():
            return False
        if force_single:
            self.rccl = RcclComm(self.device.index, RcclComm.unique_id(), 1, 0)
        else:
            if self.dist.get_backend(self.group) != "nccl":
                return False
            rank = self.dist.get_rank(self.group)
            box = [RcclComm.unique_id() if rank == 0 else None]
            self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            self.rccl = RcclComm(self.device.index, box[0], self.world, rank)
        self.loop.rccl_comm = self.rccl
        return True

    def _all_reduce(self, t):
        """In-place sum over the ranks of `t`, a slice of the statistics buffer."""
        if self.world > 1:
            if self.stream is None:
                self.ctx.sync()  # engine on its private stream: order by hand
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
                self.torch.cuda.current_stream(self.device).synchronize()
                return
            with self.torch.cuda.stream(self.stream):
                if self.dist.get_backend(self.group) != "nccl":
                    # host-staged backends (gloo dry runs) read the buffer from the host side
                    self.stream.synchronize()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def run(self, n_iters=None):
        return self.loop.run(n_iters)

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def result(self):
        """(W, H_local, (div, cost, n_iter)) as numpy."""
        return self.plan.get_w(), self.plan.get_h(), self.plan.get_objective()


def h0_columns(seed, r, t0, n_cols, T_total):
    """Columns [t0, t0 + n_cols) of the r x T_total matrix the UNSHARDED mirror draws for rand(r, n) of solve 1
    (api.sparse_nmf: RandomState(seed).random_sample((r, T_total)), src/sparse_nmf.m:112-114,:133-134), so that a sharded run
    starts from the same numbers whatever the number of ranks.  NumPy fills row by row: row k of the block is a slice of the
    k-th run of T_total draws (the stream has no jump-ahead, so every rank walks the whole stream once: host work, 80 ms at
    BASELINE config 4; api.run_basis_dnmf(..., h0="device") avoids host draws altogether)."""
    rs = np.random.RandomState(seed if seed > 0 else None)
    H0 = np.empty((int(r), int(n_cols)), order="F")
    for k in range(int(r)):
        H0[k] = rs.random_sample(int(T_total))[t0:t0 + n_cols]
    return H0


def run_basis_dnmf_sharded(Y_local, X_local, D_local, B, R_x, R_d, p, *, device=0, group=None, columns=None):
    """The 3-solve loop of run_basis_DNMF.m:36-55 with the frame axis sharded over the ranks of
    `group` (BASELINE config 4): every rank passes ITS columns of the mixture / clean / noise
    features and the replicated exemplar basis B (F x (R_x+R_d)).

      1. H-only on Y (:37-40)  -- no data exchange but the two cost scalars
      2. W-only on X with H = A_hat(1:R_x,:) (:43-47)   -- one all-reduce of the statistics / iteration
      3. W-only on D with H = A_hat(R_x+1:end,:) (:49-53)

    columns = (t0, T_total): where this rank's columns sit in the whole problem (for the initial activations of solve 1).
    Returns (B_hat [replicated, bit-identical on every rank], A_hat_local)."""
    def beta_of(p):
        cf = p.get("cf", "kl")
        return {"is": 0.0, "kl": 1.0, "ed": 2.0}.get(cf, float(p.get("beta", 1.0)))

    common = dict(beta=beta_of(p), sparsity=p.get("sparsity", 0), max_iter=int(p.get("max_iter", 100)),
                  conv_eps=float(p.get("conv_eps", 0)), cost_check=bool(p["cost_check"]), device=device, group=group)
    B = np.asarray(B, dtype=np.float64)
    r = R_x + R_d
    # rand(r, n) of solve 1 (src/sparse_nmf.m:133-134): every rank takes ITS columns [t0, t0 + T_loc) of the draw the unsharded
    # call makes for the whole T, so that B_hat does not depend on the number of GPUs (with early stops a different start
    # ends elsewhere).  `columns` = (t0, T_total) of this rank; default: an unsharded call.
    T_loc = np.asarray(Y_local).shape[1]
    t0, T_total = columns if columns is not None else (0, T_loc)
    H0 = h0_columns(int(p.get("random_seed", 1)), r, int(t0), T_loc, int(T_total))
    t1 = ShardedTrainer(Y_local, B, H0, w_update_ind=np.zeros(r, bool), h_update_ind=np.ones(r, bool), **common)
    t1.run()
    # A_hat stays on the device between the solves (round 5): p.init_h = A_hat(1:R_x,:) / A_hat(R_x+1:end,:) (:46, :52) are
    # slices of solve 1's resident H, copied device to device -- no NumPy round trip of the r x T activations
    A_dev = t1.plan.get_h_device()  # torch (T_loc, r) float32 = column-major r x T_loc
    t2 = ShardedTrainer(X_local, B[:, :R_x], A_dev[:, :R_x].contiguous(), w_update_ind=np.ones(R_x, bool),
                        h_update_ind=np.zeros(R_x, bool), **common)
    t2.run()
    B_hat_x, _, _ = t2.result()
    t3 = ShardedTrainer(D_local, B[:, R_x:r], A_dev[:, R_x:r].contiguous(), w_update_ind=np.ones(R_d, bool),
                        h_update_ind=np.zeros(R_d, bool), **common)
    t3.run()
    B_hat_d, _, _ = t3.result()
    A_hat = np.asfortranarray(A_dev.cpu().numpy().T.astype(np.float64))  # (what the caller gets back: r x T_loc)
    return np.concatenate([B_hat_x, B_hat_d], axis=1), A_hat
