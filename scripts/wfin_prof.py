"""Phase stamps of k_wfin (100 MHz real-time counter, thread 0 of every workgroup) on the Mel W-only / full / C2 loops:
diagnostic build = profiles/r06_wfin_stamps.patch applied + `python scripts/build_variant.py wfprof -DSNMF_PROF`;
SNMF_LIB_PATH=scripts/prof_build/libsnmf_wfprof.so python scripts/wfin_prof.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from se_snmf_nat_amd import Context, Plan

ctx = Context(0)
for name, F, T, r, mode in (("melw", 64, 100000, 100, "w"), ("mel", 64, 72000, 100, "full"), ("c2", 257, 100000, 256, "full"), ("tw20", 513, 72000, 20, "full")):
    rs = np.random.default_rng(1)
    V = (rs.gamma(0.5, 1.0, (F, 16)) @ rs.gamma(0.3, 1.0, (16, T)) + 1e-3).astype(np.float32)
    os.environ["SNMF_PROF_DUMP"] = "/tmp/wf_prof.bin"
    kw = dict(h_update_ind=np.zeros(r, bool)) if mode == "w" else {}
    pl = Plan(ctx, F, T, r, beta=1.0, max_iter=60, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
    pl.set_v(V); pl.set_w(rs.random((F, r))); pl.set_h(rs.random((r, T)).astype(np.float32)); pl.init()
    pl.run_async(50); ctx.sync()
    pl.close()
    x = np.fromfile("/tmp/wf_prof.bin", dtype=np.uint64).reshape(-1, 12)[:r, :8].astype(np.int64)
    t0 = x[:, 0].min()
    x = (x - t0) * 10.0 / 1000.0  # us
    names = ["start", "loads issued + stop flag", "slab items summed", "barrier", "column sums (QP)", "objective tree", "conv test", "wapply_column done"]
    print(f"{name}: {r} workgroups; first start 0, last start {x[:, 0].max():.2f} us, last end {x[:, 7].max():.2f} us")
    for i in range(1, 8):
        d = x[:, i] - x[:, i - 1]
        print(f"   {names[i]:28s} +{d.mean():6.2f} us (min {d.min():5.2f} max {d.max():5.2f})   reached at {x[:, i].mean():6.2f}")
