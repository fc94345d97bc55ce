#!/usr/bin/env python3
"""bench.py -- NMF multiplicative-update iterations/s on MI355X (BASELINE.json metric).

Workload at N=1: BASELINE.json configs[1] = single-MI355X sparse NMF basis training,
257 x 100 000 frames, r = 256, KL divergence, L1 sparsity 5 (SURVEY.md §8d "C2").  A "step" is one
full iteration of src/sparse_nmf.m:186-286: H half-step + W half-step + objective, on synthetic
|STFT|-like data that is already resident in HBM when the timed region starts.

N>1 (one rank per GPU over RCCL): the SAME total problem with the frame axis sharded across ranks (strong
scaling), one all-reduce of the W statistics per iteration (se_snmf_nat_amd/dist.py).  Launched either by
torch.distributed.run (the driver's way: RANK / WORLD_SIZE in the environment) or by this script itself: with
`--gpus N` and no WORLD_SIZE it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
BEFORE anything in this process touches HIP and relays rank 0's line.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

import numpy as np  # noqa: E402

F_, T_, R_ = 257, 100_000, 256
SPARSITY = 5.0
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA
# The chip needs ~20 ms of continuous load before it holds its steady clock (scripts/clock_ramp_probe.py: the first 10
# iterations after ANY idle gap, even 50 ms, run 19 % slow, the next 20 run 3-7 % slow, at every shard size), and a
# 5-step warm-up is 3 ms (0.6 ms at the 8-GPU shard size).  So SETTLE untimed iterations of the same loop run
# directly ahead of the W warm-up steps; the timed region is still exactly K steps.  Reported as config.settle_steps.
SETTLE = 150


def make_problem(F, T, r, t0=0, t1=None):
    """Deterministic synthetic input: V = Gamma(.5)*Gamma(.3) + 1e-9, U(0,1) inits.
    Generated in frame blocks so that every rank can draw exactly its own columns."""
    t1 = T if t1 is None else t1
    rd = np.random.default_rng(0)
    Wt = rd.gamma(0.5, 1.0, size=(F, r))
    ri = np.random.default_rng(1)
    W0 = ri.random((F, r))
    # per-block streams keyed by the block index keep shards reproducible for any world size
    blk = 1000
    Vs, Hs = [], []
    for b in range(t0 // blk, (t1 + blk - 1) // blk):
        g = np.random.default_rng([2, b])
        Ht = g.gamma(0.3, 1.0, size=(r, blk))
        H0 = g.random((r, blk))
        lo, hi = max(t0, b * blk) - b * blk, min(t1, (b + 1) * blk) - b * blk
        Vs.append((Wt @ Ht[:, lo:hi]) + 1e-9)
        Hs.append(H0[:, lo:hi])
    return np.concatenate(Vs, axis=1), W0, np.concatenate(Hs, axis=1)


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        n = [d.get("num_threads", 0) for d in threadpool_info() if d.get("user_api") == "blas"]
        return max(n) if n else os.cpu_count()
    except Exception:
        return os.cpu_count()


def cpu_baseline(F, T, r, budget_iters=8, repeats=3):
    """The reference's CPU path, represented by the fp64 oracle restatement (MATLAB is not
    available): same operation sequence as src/sparse_nmf.m including MATLAB's duplicated
    (V./Lam)*H' product, BLAS-backed.  Bounded sample: best of `repeats` runs of `budget_iters`
    iterations of the SAME 257 x 100000, r = 256 workload (best-of-3 wall time, inputs excluded)."""
    from oracle.sparse_nmf_oracle import sparse_nmf as oracle_nmf
    V, W0, H0 = make_problem(F, T, r)
    p = dict(cf="kl", sparsity=SPARSITY, max_iter=budget_iters, conv_eps=0, init_w=W0, init_h=H0, cost_check=1)
    dts = []
    ncpu = os.cpu_count()
    cap_note = ""
    try:  # "all host cores" -- ask the BLAS for every logical CPU and report what it grants
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=ncpu, user_api="blas")
    except Exception:  # noqa: BLE001
        limiter = None
    nthr = blas_threads()
    if nthr < ncpu:
        cap_note = f" (asked for {ncpu}: the OpenBLAS bundled with NumPy is built for at most {nthr} threads)"
    for _ in range(repeats):
        t = time.perf_counter()
        oracle_nmf(V, p, mimic_matlab_flops=True)
        dts.append(time.perf_counter() - t)
    if limiter is not None:
        limiter.restore_original_limits()
    dt = min(dts)
    return {"value": budget_iters / dt, "unit": "iterations/s", "cores": nthr, "kind": "port",
            "sample": f"best of {repeats} runs of {budget_iters} iterations of the same {F}x{T} r={r} KL workload, fp64 "
                      f"NumPy/OpenBLAS oracle (stand-in for MATLAB sparse_nmf.m, not MATLAB itself); {nthr} BLAS threads "
                      f"of {ncpu} logical CPUs{cap_note} (element-wise passes are single-threaded NumPy); runs took "
                      + ", ".join(f"{x:.1f}" for x in dts) + " s"}


def algorithmic_bytes(family, F, T, r):
    """SURVEY.md section 8d's per-launch HBM bytes of a half-step (fp32): the H step reads V and H and writes H
    (4(FT + 2rT)); the W statistics read V and H once (4(FT + rT); their split-T slabs are extra and not counted)."""
    return 4.0 * (F * T + 2.0 * r * T) if family == "hstep" else 4.0 * (F * T + r * T)


def csrc_digest():
    """sha256 over the kernel sources (names + bytes, sorted): what scripts/summarize_prof.py stamps into the traffic file
    on the box that measured it, so that a later bench can tell whether the kernels changed since (`traffic_stale`)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(_ROOT, "se_snmf_nat_amd", "csrc", "*"))):
        if os.path.isfile(fn) and fn.endswith((".h", ".hip", ".cpp")):
            h.update(os.path.basename(fn).encode())
            h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def pick_traffic_key(tj, kname):
    """The entry of a profiles/*_traffic.json that belongs to the launch of `kname` the timed region runs: the key must BE
    an instantiation of that kernel (`kname<...>`, never a substring match of another family); among those the one launched
    most often in the profiled command (`calls`, stamped by summarize_prof.py), else -- older files -- the objective
    variant (`<true` first template argument of the role pipelines), else the one that moved the most bytes (warm-up and
    drop-in launches of the same family run on a few thousand frames)."""
    keys = [k for k in tj if not k.startswith("_") and (k == kname or k.startswith(kname + "<"))]
    if not keys:
        return None
    if all("calls" in tj[k] for k in keys):
        return max(keys, key=lambda k: (tj[k]["calls"], tj[k]["total_bytes"]))
    obj = [k for k in keys if k.startswith(kname + "<true")]
    return max(obj or keys, key=lambda k: tj[k]["total_bytes"])


def committed_traffic(kname, flops, alg_bytes, root=_ROOT):
    """(traffic, traffic_source, frac_rocprof) from the newest committed rocprofv3 summaries of this command."""
    import csv
    import glob
    fns = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_traffic.json")))
    if not fns:
        return None, None, None
    fn = fns[-1]
    tj = json.load(open(fn))
    key = pick_traffic_key(tj, kname)
    if key is None:
        return None, None, None
    traffic = tj[key]["total_bytes"]
    stamp = tj.get("_csrc_digest")
    src = {"file": "profiles/" + os.path.basename(fn), "kernel": key, "commit": tj.get("_source_commit"),
           "csrc_digest": stamp, "traffic_stale": (stamp != csrc_digest()) if stamp else None,
           "note": "quoted from the committed rocprofv3 --pmc passes of this command (a profiler cannot run inside bench.py); "
                   "traffic_stale = a file under se_snmf_nat_amd/csrc changed since those passes (null: the file predates the stamp)"}
    frac = None
    ks = fn.replace("_traffic.json", "_kernel_stats.csv")
    if os.path.exists(ks):
        for row in csv.reader(ln for ln in open(ks) if not ln.startswith("#")):
            if row and row[0] == key:
                avg_ns = float(row[3])
                frac = flops / (avg_ns * 1e-9) / 1e12 / PEAK_F32_MFMA_TFLOPS
                src["rocprof_avg_ms"] = avg_ns * 1e-6
                src["rocprof_calls"] = int(row[1])
    return traffic, src, frac


def cost_vs_oracle(F, T, r, n_it, final_cost):
    """|final_cost - oracle cost after the same number of iterations| / oracle cost, from the committed full-size
    golden (tests/golden/make_golden_c2.py: the fp64 oracle on exactly make_problem's inputs).  None when the run is
    not the golden's configuration or went past its horizon."""
    fn = os.path.join(_ROOT, "tests", "golden", "c2_full_257x100000_r256.npz")
    if final_cost is None or not os.path.exists(fn):
        return None
    g = np.load(fn)
    if (int(g["F"]), int(g["T"]), int(g["r"])) != (F, T, r) or float(g["sparsity"]) != SPARSITY:
        return None
    if not (1 <= n_it <= len(g["cost"])):
        return None
    ref = float(g["cost"][n_it - 1])
    return {"iterations": int(n_it), "oracle_cost": ref, "rel_diff": abs(final_cost - ref) / ref}


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as children (torch.distributed.run, one process per GPU) and
    relay rank 0's JSON line.  Nothing in THIS process has touched HIP (no torch.cuda call, no libsnmf call)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--dim-F", str(args.F), "--dim-T", str(args.T), "--dim-r", str(args.r), "--c5-T", str(args.c5_T), "--c4-T", str(args.c4_T)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in pr.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if pr.returncode != 0 or line is None:
        sys.stderr.write(pr.stdout)
        sys.exit(pr.returncode or 1)
    print(line)


def oneshot_child(args):
    """The SAME sharded problem through the one-process multi-GPU entry of the C ABI (snmf_multi_*: one host thread per
    device, the one-shot peer-store all-reduce of csrc/snmf_multi.h) -- the exchange SURVEY.md section 5 / 8e asks to be
    reported next to RCCL.  Runs in a process of its own that rank 0 starts BEFORE it touches HIP and releases (one line
    on stdin) after the RCCL leg; prints one JSON object."""
    if sys.stdin.readline().strip() != "go":  # wait for rank 0's go (a closed pipe = rank 0 is gone: nothing to do)
        return
    out = {}
    try:
        import ctypes as C
        from se_snmf_nat_amd import _lib
        from se_snmf_nat_amd.api import _make_params
        lib = _lib.load()
        N, K, Wm = args.gpus, args.steps, args.warmup
        F, T, r = args.F, args.T, args.r
        if "SNMF_FORCE_DEVICE" in os.environ:
            devs = np.full(N, int(os.environ["SNMF_FORCE_DEVICE"]), np.int32)  # single-GPU dry run: the ranks share it
        else:
            devs = np.arange(N, dtype=np.int32)
        V, W0, H0 = make_problem(F, T, r)
        Vf, Hf, Wf = np.asfortranarray(V, np.float32), np.asfortranarray(H0, np.float32), np.asfortranarray(W0)
        n_pre = SETTLE + 2 * Wm  # as many untimed iterations as the RCCL leg runs ahead of its timed region
        sp = _make_params(F, T, r, 1.0, n_pre + K + 1, 0.0, 1, True, 0, SPARSITY, None, None)
        h = C.c_void_p()
        _lib.check(lib.snmf_multi_create(C.c_void_p(devs.ctypes.data), N, C.byref(sp), None, C.byref(h)))
        try:
            _lib.check(lib.snmf_multi_set_v_f32(h, C.c_void_p(Vf.ctypes.data), F))
            _lib.check(lib.snmf_multi_set_w_f64(h, C.c_void_p(Wf.ctypes.data), F))
            _lib.check(lib.snmf_multi_set_h_f32(h, C.c_void_p(Hf.ctypes.data), r))
            _lib.check(lib.snmf_multi_init(h))
            done = C.c_int32()
            _lib.check(lib.snmf_multi_run(h, n_pre, C.byref(done)))
            t = time.perf_counter()
            _lib.check(lib.snmf_multi_run(h, K, C.byref(done)))  # returns after every rank's stream has drained
            dt = time.perf_counter() - t
            div, cost = np.zeros(n_pre + K + 1), np.zeros(n_pre + K + 1)
            n = C.c_int32()
            _lib.check(lib.snmf_multi_get_objective(h, C.c_void_p(div.ctypes.data), C.c_void_p(cost.ctypes.data), C.byref(n)))
        finally:
            lib.snmf_multi_destroy(h)
        last = [c for c in cost if c != 0.0]
        out = {"ms_per_step": dt / K * 1e3, "value": K / dt, "unit": "iterations/s", "devices": [int(d) for d in devs],
               "ordering": "device-side arrival flags" if len(set(int(d) for d in devs)) == N else "hipEvents + host barrier (ranks share a device)",
               "entry": "snmf_multi_* (one process, one host thread per device, peer-store all-reduce)",
               "steps": K, "final_cost": float(last[-1]) if last else None}
        out["cost_vs_oracle"] = cost_vs_oracle(F, T, r, len(last), out["final_cost"])
    except Exception as e:  # noqa: BLE001 -- the extra leg must never fail the bench
        out = {"error": f"{type(e).__name__}: {e}"}
    print("ONESHOT " + json.dumps(out), flush=True)


def start_oneshot_child(args):
    """Started by rank 0 before anything in it touches HIP (a process that has initialised the GPU must not fork + exec)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--oneshot-child", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--dim-F", str(args.F), "--dim-T", str(args.T), "--dim-r", str(args.r)]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    try:
        return subprocess.Popen(cmd, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except Exception:  # noqa: BLE001
        return None


def finish_oneshot_child(proc, limit_s=420):
    if proc is None:
        return {"error": "the one-shot child process could not be started"}
    try:
        outs, errs = proc.communicate("go\n", timeout=limit_s)
    except Exception as e:  # noqa: BLE001
        proc.kill()
        return {"error": f"one-shot leg: {type(e).__name__}: {e}"}
    for ln in outs.splitlines():
        if ln.startswith("ONESHOT "):
            return json.loads(ln[len("ONESHOT "):])
    return {"error": f"one-shot leg exited with code {proc.returncode}: {errs.strip()[-400:]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the legs behind the timed region -- the host-array drop-in call and the solves from a random start -- (profiling "
                         "passes: their launches -- fresh plans, first iterations -- would be averaged into the timed loop's kernel statistics)")
    ap.add_argument("--oneshot-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--c5-T", dest="c5_T", type=int, default=500_000, help=argparse.SUPPRESS)  # frames of the extra C5 leg (tests shrink it)
    ap.add_argument("--c4-T", dest="c4_T", type=int, default=800_000, help=argparse.SUPPRESS)  # frames IN ALL of the extra C4 leg
    # --dim-*: the spellings self_launch() passes on (torch.distributed.run's own parser prefix-matches a bare --r)
    ap.add_argument("--F", "--dim-F", dest="F", type=int, default=F_)
    ap.add_argument("--T", "--dim-T", dest="T", type=int, default=T_)
    ap.add_argument("--r", "--dim-r", dest="r", type=int, default=R_)
    args = ap.parse_args()
    F, T, r = args.F, args.T, args.r
    K, W = args.steps, args.warmup
    if args.oneshot_child:
        return oneshot_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    flops_half = 4.0 * F * T * r  # per launch of k_hstep or k_wstats (2 contractions each), whole problem

    if world == 1 and args.gpus == 1:
        # single GPU: plain C-ABI path, no torch needed
        from se_snmf_nat_amd import Context, Plan
        ctx = Context(0)
        V, W0, H0 = make_problem(F, T, r)
        plans = []
        for _ in range(2):  # [0]: per-kernel HIP-event pass, [1]: the headline pass (events add a little host work)
            pl = Plan(ctx, F, T, r, beta=1.0, max_iter=SETTLE + W + K + 1, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
            pl.set_v(V.astype(np.float32))
            pl.set_w(W0)
            pl.set_h(H0.astype(np.float32))
            pl.init()
            plans.append(pl)
        plan, plan2 = plans
        desc = plan.describe()
        plan.run_async(SETTLE + W)
        ctx.sync()
        ctx.timing(True)
        plan.run_async(K)
        ctx.sync()
        fam = {f: ctx.timing_get(f) for f in ("hstep", "wstats", "reduce", "wapply", "wfin")}
        ctx.timing(False)
        plan2.run_async(SETTLE + W)
        ctx.sync()
        t = time.perf_counter()
        plan2.run_async(K)
        ctx.sync()
        dt = time.perf_counter() - t
        div, cost, n_it = plan2.get_objective()
        ms = dt / K * 1e3
        dom = max(("hstep", "wstats"), key=lambda f: fam[f][0])
        # the kernel(s) actually launched for the dominant half-step (from the plan's own description)
        if dom == "hstep":
            kname = next((k for k in ("k_hstep_rh", "k_hstep_rp", "k_hstep_m", "k_iter_sf", "k_hstep_sf") if k in desc), "k_hstep")
        else:
            kname = "k_wstats_sf" if "k_wstats_sf" in desc else "k_wstats"
        # HBM bytes per launch of the dominant kernel: from the SEPARATE rocprofv3 --pmc passes of the
        # same command (scripts/prof.sh -> scripts/summarize_prof.py -> profiles/*_traffic.json);
        # a profiler cannot run inside this process, so the committed measurement is quoted.
        alg_bytes = algorithmic_bytes(dom, F, T, r)
        traffic, traffic_source, frac_rocprof = None, None, None
        if (F, T, r) == (F_, T_, R_):
            traffic, traffic_source, frac_rocprof = committed_traffic(kname, flops_half, alg_bytes)
        last_cost = [c for c in cost if c != 0.0]
        ach = flops_half / (fam[dom][0] * 1e-3) / 1e12 if fam[dom][0] > 0 else 0.0
        out = {
            "metric": "NMF multiplicative-update iterations/sec (FxTxr)", "value": K / dt, "unit": "iterations/s",
            "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"single-MI355X sparse NMF basis train (BASELINE configs[1]): {F}x{T} frames, "
                                   f"r={r}, KL, sparsity={SPARSITY}, full W+H update + objective per step",
                       "F": F, "T": T, "r": r, "beta": 1, "settle_steps": SETTLE, "geometry": desc},
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE/WRITE_SIZE passes)",
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_ratio": (traffic / alg_bytes) if traffic else None,
                         "frac_rocprof": frac_rocprof,
                         "algorithmic_flops_per_launch": flops_half,
                         "kernel_ms": {f: fam[f][0] for f in fam if fam[f][1]}, "launches": {f: fam[f][1] for f in fam if fam[f][1]},
                         "kernel_ms_note": "HIP events on the engine's stream around every launch of a family, from a SEPARATE "
                                           "event-instrumented pass of the same loop (the events add host work, so the "
                                           "averages need not sum to ms_per_step, which comes from the un-instrumented pass); "
                                           "wfin = k_wfin, the chunk reduction and the W update in one launch (the step API "
                                           "and the multi-GPU loops run them as k_reduce + k_wapply)",
                         "whole_iteration_TFLOPs": 2 * flops_half / (ms * 1e-3) / 1e12},
            "final_cost": float(last_cost[-1]) if last_cost else None,
        }
        out["cost_vs_oracle"] = cost_vs_oracle(F, T, r, len(last_cost), out["final_cost"])
        # The same resident loop FROM A RANDOM START (no settle steps): kernel time depends on the iterate (dense random
        # activations toggle more operand bits than the sparse ones the L1 penalty leaves; the clock follows the power), so
        # a short solve pays more per iteration than `value`.  200 iterations = BASELINE configs[1]; 100 = the reference's
        # default max_iter (settings/initial_setting_SNMF_NAT.m:108).  Outside the timed region; never `value`.
        try:
            if args.no_dropin:
                raise RuntimeError("skipped (--no-dropin: profiling pass)")
            frs = {}
            for n_it in (200, 100):
                best = None
                for _ in range(2):
                    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=n_it, conv_eps=0.0, cost_check=True, sparsity=SPARSITY)
                    pl.set_v(V.astype(np.float32))
                    pl.set_w(W0)
                    pl.set_h(H0.astype(np.float32))
                    pl.init()
                    ctx.sync()
                    t = time.perf_counter()
                    pl.run_async(n_it)
                    ctx.sync()
                    dts = time.perf_counter() - t
                    pl.close()
                    best = dts if best is None else min(best, dts)
                frs[f"iters_{n_it}"] = {"iterations_per_s": n_it / best, "ms_per_step": best / n_it * 1e3, "solve_ms": best * 1e3}
            out["from_random_start"] = frs["iters_200"]["iterations_per_s"]
            out["from_random_start_detail"] = dict(frs, note="one resident solve from the random start, data in HBM, no settle steps; "
                                                             "includes the final objective pass; best of two")
        except Exception as e:  # noqa: BLE001 -- the extra leg must never fail the bench
            out["from_random_start"] = None
            out["from_random_start_detail"] = {"error": f"{type(e).__name__}: {e}"}
        # What a caller of the drop-in boundary waits for (outside the timed region; never `value`): the same K iterations
        # through [w, h, objective] = sparse_nmf(v, p) on HOST fp64 arrays -- MATLAB's doubles in, W / H / objective out
        # (snmf_sparse_nmf_f64: chunked pinned upload, solve, download).  scripts/bench_dropin.py has the full breakdown.
        for pl in plans:
            pl.close()
        try:
            if args.no_dropin:
                raise RuntimeError("skipped (--no-dropin)")
            from se_snmf_nat_amd import sparse_nmf
            Vh, Wh, Hh = (np.asfortranarray(M, dtype=np.float64) for M in (V, W0, H0))
            pd = dict(cf="kl", sparsity=SPARSITY, max_iter=K, conv_eps=0, cost_check=1, init_w=Wh, init_h=Hh)
            sparse_nmf(Vh[:, :4096], dict(pd, init_h=Hh[:, :4096], max_iter=2), ctx=ctx)  # pinned buffers exist, code objects loaded
            best = None
            for _ in range(4):  # (the first full-size call of a process pays first-touch costs on the result arrays: 48 against 34 ms)
                ctx.xfer_stats(reset=True)
                t = time.perf_counter()
                sparse_nmf(Vh, pd, ctx=ctx)
                dtd = time.perf_counter() - t
                st = ctx.xfer_stats()
                if best is None or dtd < best[0]:
                    best = (dtd, st)
            dtd, st = best
            out["dropin_ms"] = dtd * 1e3
            out["dropin"] = {"call": "sparse_nmf(v, p): host fp64 arrays in, w / h / objective out (snmf_sparse_nmf_f64)", "iterations": K,
                             "call_ms": dtd * 1e3, "resident_ms": ms * K, "h2d_ms": st["h2d_wall_s"] * 1e3, "h2d_MB": st["h2d_bytes"] / 1e6,
                             "d2h_ms": st["d2h_wall_s"] * 1e3, "d2h_MB": st["d2h_bytes"] / 1e6,
                             "note": "includes plan creation / destruction and the mirror's copies of the in/out arrays"}
        except Exception as e:  # noqa: BLE001 -- the extra leg must never fail the bench
            out["dropin_ms"] = None
            out["dropin"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F, T, r)
        print(json.dumps(out))
        return

    # ---- multi-GPU: one rank per GPU, frames sharded, RCCL all-reduce of the W statistics ----
    oneshot = start_oneshot_child(args) if (rank == 0 and world > 1) else None  # (before this process touches HIP)
    import torch
    import torch.distributed as dist
    from se_snmf_nat_amd.dist import ShardedTrainer, shard_bounds
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # SNMF_DIST_BACKEND / SNMF_FORCE_DEVICE exist for single-GPU dry runs of the multi-rank path
    # (two ranks on one device over gloo); the driver's launch uses RCCL, one rank per GPU.
    backend = os.environ.get("SNMF_DIST_BACKEND", "nccl")
    if "SNMF_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SNMF_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    t0, t1 = shard_bounds(T, world, rank)
    V, W0, H0 = make_problem(F, T, r, t0, t1)
    tr = ShardedTrainer(V.astype(np.float32), W0, H0.astype(np.float32), beta=1.0, sparsity=SPARSITY,
                        max_iter=SETTLE + 2 * W + K + 1, conv_eps=0.0, cost_check=True, device=local_rank)
    desc = tr.plan.describe()
    # per-kernel HIP-event pass (own, untimed): the events add host work, so the headline pass below runs without them
    tr.run(SETTLE)
    tr.sync()
    tr.ctx.timing(True)
    tr.run(W)
    tr.sync()
    fam = {f: tr.ctx.timing_get(f)[0] for f in ("hstep", "wstats", "reduce", "wapply", "wfin")}
    tr.ctx.timing(False)
    tr.run(W)  # the contract's W warm-up steps, directly ahead of the timed region
    tr.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.run(K)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    dtt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dtt, op=dist.ReduceOp.MAX)
    dt = float(dtt.item())
    div, cost, n_it = tr.plan.get_objective()
    last_cost = [c for c in cost if c != 0.0]
    fams = [fam]
    replicas_identical = None
    if world > 1:
        fams = [None] * world
        dist.all_gather_object(fams, fam)
        # every rank applies the same deterministic W update to the same summed statistics: the replicas of W must agree BIT FOR BIT
        # (outside the timed region; on a node with a device per rank this is the check that the exchange delivered the same sum to all)
        import zlib
        crc = torch.tensor([float(zlib.crc32(np.ascontiguousarray(tr.plan.get_w()).tobytes()))], dtype=torch.float64, device="cuda")
        lo, hi = crc.clone(), crc.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(lo.item() == hi.item())
    if rank == 0:
        ms = dt / K * 1e3
        tot = 2 * flops_half / (ms * 1e-3) / 1e12
        out = {
            "metric": "NMF multiplicative-update iterations/sec (FxTxr)", "value": K / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{world}xMI355X frame-sharded sparse NMF basis train: {F}x{T} frames total, r={r}, "
                                   f"KL, sparsity={SPARSITY}, one {backend} all-reduce of the W statistics per step",
                       "F": F, "T": T, "r": r, "beta": 1, "parallelism": f"frames/{world}", "settle_steps": SETTLE + W,
                       "geometry": desc},
            "roofline": {"bound": "mfma", "kernel": "whole iteration (all ranks)", "achieved": tot,
                         "peak": PEAK_F32_MFMA_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": tot / (PEAK_F32_MFMA_TFLOPS * world), "traffic": None,
                         "kernel_ms_per_rank": fams},
            "final_cost": float(last_cost[-1]) if last_cost else None,
        }
        if replicas_identical is not None:
            out["w_replicas_bit_identical"] = replicas_identical
        out["cost_vs_oracle"] = cost_vs_oracle(F, T, r, len(last_cost), out["final_cost"])
        # the contract times the CPU baseline on rank 0 at N = 1 only
    del tr
    torch.cuda.empty_cache()
    guard = None
    if world > 1:
        # The extra legs below must never cost the headline measurement: if a rank fails INSIDE one of their loops its peers sit in
        # a collective until the backend's own time-out (ten minutes) kills the job -- and the line with it.  A timer on every rank
        # ends the process first; rank 0 prints what it has.
        import threading

        def bail():
            if rank == 0:
                out.setdefault("c5_strong", {"error": "the extra legs timed out"})
                out.setdefault("c4_dnmf", {"error": "the extra legs timed out"})
                out.setdefault("exchange_oneshot", {"error": "skipped: the extra legs timed out"})
                print(json.dumps(out), flush=True)
            os._exit(0)

        guard = threading.Timer(float(os.environ.get("SNMF_BENCH_EXTRA_TIMEOUT", "420")), bail)
        guard.daemon = True
        guard.start()
    if world > 1:
        # SURVEY.md section 8d asks strong scaling on "C2 and C5": the SAME loop on BASELINE configs[4] (513 x 500000, r = 512,
        # beta = 2, lambda = 50: 11 ms per iteration on one GPU, the config where the fixed costs of an iteration are small
        # beside its compute).  Reported beside `value`, never instead of it; a failure of this extra leg never fails the bench.
        c5 = c5_strong_leg(world, rank, local_rank, torch, dist, T=args.c5_T)
        if rank == 0:
            out["c5_strong"] = c5
        # BASELINE configs[3], the config north_star's ">= 6x at 8 GPUs" names: the sharded basis-training path run_basis_DNMF.m:36-55
        # (3 solves x 50 iterations, 513 x 800000 frames in all, R_x = R_d = 100), A_hat resident between the solves; strong scaling,
        # with the one-GPU time of the same problem measured by rank 0 in the same process
        c4 = c4_dnmf_leg(world, rank, local_rank, torch, dist, T_total=args.c4_T)
        if rank == 0:
            out["c4_dnmf"] = c4
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1:
            # the same problem through the one-shot peer-store exchange behind the C ABI, AFTER the timed RCCL leg (the
            # other ranks are gone or idle by now); reported beside `value`, never instead of it
            out["exchange_oneshot"] = finish_oneshot_child(oneshot)
        if guard:
            guard.cancel()
        print(json.dumps(out), flush=True)
    elif guard:
        guard.cancel()


def c5_strong_leg(world, rank, local_rank, torch, dist, F=513, T=500_000, r=512, steps=10, warm=4):
    """BASELINE configs[4] frame-sharded over the ranks (strong scaling), timed like the headline leg: barrier + synchronize on
    both sides, MAX over ranks.  Every rank reports whether its set-up worked BEFORE anybody enters the loop's collectives."""
    from se_snmf_nat_amd.dist import ShardedTrainer, shard_bounds
    tr, err = None, None
    try:
        t0, t1 = shard_bounds(T, world, rank)
        g = np.random.default_rng([5, rank])
        Wt = np.random.default_rng(5).gamma(0.5, 1.0, size=(F, 32)).astype(np.float32)
        V = np.empty((F, t1 - t0), np.float32, order="F")
        H0 = np.empty((r, t1 - t0), np.float32, order="F")
        for a in range(0, t1 - t0, 50000):  # in blocks: 513 x 500000 floats are 1 GB
            b = min(t1 - t0, a + 50000)
            V[:, a:b] = Wt @ g.gamma(0.3, 1.0, size=(32, b - a)).astype(np.float32) + 1e-9
            H0[:, a:b] = g.random((r, b - a), dtype=np.float32)
        W0 = np.random.default_rng(6).random((F, r))
        tr = ShardedTrainer(V, W0, H0, beta=2.0, sparsity=50.0, max_iter=warm + steps + 1, conv_eps=0.0, cost_check=True, device=local_rank)
        del V, H0
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 1.0:
        return {"error": err or "another rank could not set the C5 shard up"}
    # (a rank that raises INSIDE the loop leaves its peers in the loop's collective until the process group's time-out: nothing
    #  short of that can release them; what CAN be guaranteed is that no rank reports a number unless every rank finished)
    try:
        tr.run(warm)
        tr.sync()
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        tr.run(steps)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        fin = torch.tensor([1.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(fin, op=dist.ReduceOp.MIN)
        if float(fin.item()) < 1.0:
            return {"error": "a rank did not finish the C5 loop"}
        _d, cost, _n = tr.plan.get_objective()
        last = [c for c in cost if c != 0.0]
        desc = tr.plan.describe()
        del tr
        torch.cuda.empty_cache()
        # executed flop per iteration on this path: H step 6 F T r, W step V*H' 2 F T r + Gram 2 r^2 T (+ W*Gram, small)
        fl = 8.0 * F * T * r + 2.0 * r * r * T
        return {"workload": f"{world}xMI355X frame-sharded BASELINE configs[4]: {F}x{T} r={r} beta=2 lambda=50", "ms_per_step": dt / steps * 1e3,
                "value": steps / dt, "unit": "iterations/s", "steps": steps, "warmup": warm, "scaling": "strong",
                "executed_TFLOPs": fl * steps / dt / 1e12, "frac_of_peak": fl * steps / dt / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
                "final_cost": float(last[-1]) if last else None, "geometry": desc}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def c4_dnmf_leg(world, rank, local_rank, torch, dist, F=513, T_total=800_000, R_x=100, R_d=100, iters=50):
    """BASELINE configs[3] on the process-per-GPU path: run_basis_DNMF.m:36-55 -- H-only on Y (r = R_x + R_d), W-only on X and on D
    with the activations of solve 1 -- with the T_total frames of the corpus sharded over the ranks (STRONG scaling: the problem is
    the same at every N), one all-reduce of the W statistics per iteration of solves 2 / 3, A_hat handed from solve 1 to solves
    2 / 3 on the device.  The three feature blocks are resident (torch CUDA tensors) when the timed region starts; plan creation
    is inside it.  Timed like the headline leg: barrier + synchronize on both sides, MAX over ranks.  Then rank 0 ALONE runs the
    whole T_total problem through the same code (a one-rank process group) while the others wait: `seconds_one_gpu`, so that the
    line carries its own strong-scaling ratio, measured in one process on one box."""
    from se_snmf_nat_amd.dist import ShardedTrainer
    err, dev = None, torch.device("cuda", local_rank)
    r, BLK = R_x + R_d, 50_000
    Wx = np.random.default_rng(41).gamma(0.5, 1.0, size=(F, 24)).astype(np.float32)
    Wd = np.random.default_rng(42).gamma(0.5, 1.0, size=(F, 16)).astype(np.float32)
    B = np.random.default_rng(43).random((F, r)) + 0.05

    def frames(t0, t1):
        """columns [t0, t1) of the synthetic corpus as device tensors (T, F) / (T, r): generated in global blocks of BLK columns, so
        that every world size sees the same matrices"""
        Xd = torch.empty((t1 - t0, F), dtype=torch.float32, device=dev)
        Dd = torch.empty((t1 - t0, F), dtype=torch.float32, device=dev)
        H0 = torch.empty((t1 - t0, r), dtype=torch.float32, device=dev)
        for blk in range(t0 // BLK, (t1 + BLK - 1) // BLK):
            g = np.random.default_rng([4, blk])
            n = min(T_total, (blk + 1) * BLK) - blk * BLK
            xb = (Wx @ g.gamma(0.3, 1.0, size=(24, n)).astype(np.float32) + 1e-9).T
            db = (Wd @ g.gamma(0.3, 1.0, size=(16, n)).astype(np.float32) + 1e-9).T
            hb = g.random((n, r), dtype=np.float32)
            lo, hi = max(t0, blk * BLK), min(t1, blk * BLK + n)
            sl = slice(lo - blk * BLK, hi - blk * BLK)
            Xd[lo - t0:hi - t0] = torch.from_numpy(np.ascontiguousarray(xb[sl])).to(dev)
            Dd[lo - t0:hi - t0] = torch.from_numpy(np.ascontiguousarray(db[sl])).to(dev)
            H0[lo - t0:hi - t0] = torch.from_numpy(np.ascontiguousarray(hb[sl])).to(dev)
        return Xd + Dd, Xd, Dd, H0

    try:
        t0, t1 = T_total * rank // world, T_total * (rank + 1) // world
        Yd, Xd, Dd, H0 = frames(t0, t1)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 1.0:
        return {"error": err or "another rank could not set the C4 shard up"}
    solo = dist.new_group([0]) if world > 1 else None  # (every rank has to make this call)
    try:
        common = dict(beta=1.0, sparsity=5.0, max_iter=iters, conv_eps=0.0, cost_check=True, device=local_rank)

        def three_solves(Yd, Xd, Dd, H0, group=None):
            t1 = ShardedTrainer(Yd, B, H0, w_update_ind=np.zeros(r, bool), h_update_ind=np.ones(r, bool), group=group, **common)
            t1.run()
            A = t1.plan.get_h_device()
            t2 = ShardedTrainer(Xd, B[:, :R_x], A[:, :R_x].contiguous(), w_update_ind=np.ones(R_x, bool), h_update_ind=np.zeros(R_x, bool),
                                group=group, **common)
            t2.run()
            t3 = ShardedTrainer(Dd, B[:, R_x:], A[:, R_x:].contiguous(), w_update_ind=np.ones(R_d, bool), h_update_ind=np.zeros(R_d, bool),
                                group=group, **common)
            t3.run()
            t3.sync()
            c3 = [c for c in t3.plan.get_objective()[1] if c != 0.0]
            return float(c3[-1]) if c3 else None, t1.plan.describe(), t2.plan.describe()

        three_solves(Yd, Xd, Dd, H0)  # warm-up: code objects, cached device blocks, the clock
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        cost3, d1, d2 = three_solves(Yd, Xd, Dd, H0)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        fin = torch.tensor([1.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(fin, op=dist.ReduceOp.MIN)
        if float(fin.item()) < 1.0:
            return {"error": "a rank did not finish the C4 loop"}
        fl = iters * (4.0 * F * T_total * r + 2 * 4.0 * F * T_total * R_x)  # H-only: Lam + contraction at r; W-only (x2): the same at R_x (= R_d)
        out = {"workload": f"{world}xMI355X run_basis_DNMF (BASELINE configs[3]): 513 x {T_total} frames sharded over the ranks, "
                           f"R_x = R_d = {R_x}, 3 solves x {iters} iterations, A_hat resident between the solves", "seconds": dt,
               "value": 3 * iters / dt, "unit": "solver iterations/s", "scaling": "strong", "frames_total": T_total,
               "algorithmic_TFLOPs": fl / dt / 1e12, "frac_of_peak": fl / dt / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
               "final_cost_solve3": cost3, "geometry_solve1": d1, "geometry_solve2": d2}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    if world > 1:
        # the whole problem on rank 0's GPU alone, same code, a one-rank group; a failure here (memory) leaves the sharded numbers alone
        one = None
        try:
            del Yd, Xd, Dd, H0
            torch.cuda.empty_cache()
            if rank == 0:
                Yd, Xd, Dd, H0 = frames(0, T_total)
                three_solves(Yd, Xd, Dd, H0, group=solo)
                torch.cuda.synchronize()
                t = time.perf_counter()
                c1, _, _ = three_solves(Yd, Xd, Dd, H0, group=solo)
                torch.cuda.synchronize()
                one = (time.perf_counter() - t, c1)
        except Exception as e:  # noqa: BLE001
            out["one_gpu_error"] = f"{type(e).__name__}: {e}"
        dist.barrier()
        if one:
            out.update({"seconds_one_gpu": one[0], "strong_scaling_vs_one_gpu": one[0] / dt, "final_cost_solve3_one_gpu": one[1]})
    return out


if __name__ == "__main__":
    main()
