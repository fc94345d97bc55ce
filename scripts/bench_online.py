#!/usr/bin/env python3
"""BASELINE config 3 end to end: the online separation loop (src/NTF_sep_event_RT.m frame loop +
src/bnmf_sep_event_RT_IS16.m, shipped settings, noise-dictionary adaptation on) over the committed 1.2 s
audio fixture tiled to --seconds, shipped dictionaries.  Reports frames/s and the real-time factor
(10 ms hop), whole file in one process() call and hop-by-hop calls (real-time use), next to the CPU
oracle on a bounded sample.  One JSON line.  Usage: python scripts/bench_online.py [--seconds 12] [--cpu]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from se_snmf_nat_amd import Context  # noqa: E402
from se_snmf_nat_amd.online import OnlineSeparator, default_settings  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--no-adapt", action="store_true")
a = ap.parse_args()

G = os.path.join(ROOT, "tests", "golden")
B = np.load(os.path.join(G, "ref_data.npz"))["B"].astype(np.float64)
s = np.load(os.path.join(G, "frontend_audio.npz"))["samples"]
s = np.tile(s, int(np.ceil(a.seconds * 16000 / len(s))))[:int(a.seconds * 16000)]
p = default_settings()
if a.no_adapt:
    p["adapt_train_N"] = 0
rs = np.random.RandomState(1)
H0, Ad0 = rs.random_sample(200), rs.random_sample((50, 100))
ctx = Context(0)


def run(chunk):
    sep = OnlineSeparator(B[:, :100], B[:, 100:], p, H0=H0, Ad_blk0=Ad0, ctx=ctx)
    sep.process(s[:1600])  # warm-up: kernels loaded, buffers sized
    sep.close()
    sep = OnlineSeparator(B[:, :100], B[:, 100:], p, H0=H0, Ad_blk0=Ad0, ctx=ctx)
    t = time.perf_counter()
    if chunk is None:
        sep.process(s, flush=True)
    else:
        for i in range(0, len(s), chunk):
            sep.process(s[i:i + chunk])
        sep.process(s[:0], flush=True)
    dt = time.perf_counter() - t
    tr = sep.trace()
    sep.close()
    return dt, tr


dt_file, tr = run(None)
dt_hop, _ = run(160)
nfr = len(tr)
out = {"config": "C3 online separation end to end (bnmf_sep_event_RT_IS16, shipped settings%s), 513 bins, r=200, %d frames"
                 % (", adaptation off" if a.no_adapt else "", nfr),
       "value": nfr / dt_file, "unit": "frames/s (whole file per call)", "realtime_factor": (nfr * 0.010) / dt_file,
       "hop_by_hop_frames_per_s": nfr / dt_hop, "hop_by_hop_ms_per_frame": dt_hop / nfr * 1e3,
       "frame_solve_iters_mean": float(np.mean([t["n_iter"] for t in tr])),
       "adaptation_solves": int(sum(t["solved"] for t in tr)),
       "adaptation_iters_mean": float(np.mean([t["adapt_iters"] for t in tr if t["solved"]] or [0]))}
if a.cpu:
    from oracle.online_oracle import default_params, ntf_sep_event_rt
    po = default_params()
    if a.no_adapt:
        po["adapt_train_N"] = 0
    n = 160 * 200
    t = time.perf_counter()
    _, _, _, tro = ntf_sep_event_rt(s[:n], B[:, :100], B[:, 100:], po, H0, Ad0, return_trace=True)
    dtc = time.perf_counter() - t
    out["cpu_oracle_frames_per_s"] = len(tro) / dtc
    out["cpu_oracle_sample"] = "first %d frames, fp64 NumPy oracle, %s host threads" % (len(tro), os.cpu_count())
print(json.dumps(out), flush=True)
