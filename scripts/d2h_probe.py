#!/usr/bin/env python3
"""Where the device -> host leg of a drop-in call spends its time (csrc/snmf_tu_xfer.hip: xfer_unpack_out): Plan.get_h of the
C2 / a11 / Mel activations into a FRESH fp64 array (first touch of every page inside the call: what MATLAB's mxCreateDoubleMatrix
or np.empty hands over) against an array whose pages exist, fp64 against fp32 results."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from se_snmf_nat_amd import Context, Plan, _lib

ctx = Context(0)
lib = _lib.load()
for F, r, T in ((257, 256, 100_000), (513, 100, 72_000), (64, 100, 72_000)):
    pl = Plan(ctx, F, T, r, max_iter=1, sparsity=1.0, cost_check=True)
    pl.set_h_random(3)
    ctx.sync()
    for dt in (np.float64, np.float32):
        fn = lib.snmf_plan_get_h_f64 if dt == np.float64 else lib.snmf_plan_get_h_f32
        res = {}
        for mode in ("fresh", "touched"):
            best = 1e9
            for rep in range(4):
                out = np.empty((r, T), dtype=dt, order="F")
                if mode == "touched":
                    out[:] = 0
                ctx.xfer_stats(reset=True)
                t = time.perf_counter()
                _lib.check(fn(pl._h, C.c_void_p(out.ctypes.data), r, 0))
                d = time.perf_counter() - t
                if d < best:
                    best, st = d, ctx.xfer_stats()
            res[mode] = (best, st)
        mb = r * T * np.dtype(dt).itemsize / 1e6
        print(f"r={r} T={T} {np.dtype(dt).name}: {mb:.0f} MB out | fresh {res['fresh'][0]*1e3:.2f} ms = {mb/1e3/res['fresh'][0]:.1f} GB/s (host copy {res['fresh'][1]['d2h_host_copy_s']*1e3:.2f} ms)"
              f" | touched {res['touched'][0]*1e3:.2f} ms = {mb/1e3/res['touched'][0]:.1f} GB/s (host copy {res['touched'][1]['d2h_host_copy_s']*1e3:.2f} ms)", flush=True)
    pl.close()
