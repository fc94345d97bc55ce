"""Replay ONE fuzz case under kernel-family switches: python scripts/fuzz_case_probe2.py seed focus idx"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fuzz_cases  # noqa: E402
import se_snmf_nat_amd  # noqa: E402
from oracle.sparse_nmf_oracle import sparse_nmf as onmf  # noqa: E402

seed, focus, idx = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
calls = []
orig = se_snmf_nat_amd.sparse_nmf


def fake(V, p, **kw):
    calls.append((V, p))
    raise se_snmf_nat_amd.SnmfError(0, "skipped")


se_snmf_nat_amd.sparse_nmf = fake
fz = fuzz_cases.Fuzz(seed, focus)
for ci in range(idx + 1):
    fz.case(ci, log=lambda s: None)
se_snmf_nat_amd.sparse_nmf = orig
V, p = calls[idx]
print({k: (v if np.isscalar(v) or isinstance(v, str) else getattr(v, "shape", v)) for k, v in p.items()}, V.shape, V.min(), V.max(), flush=True)
for it in (1, 2, p["max_iter"]):
    q = dict(p, max_iter=it)
    wr, hr, orf = onmf(V, q)
    for hs, ws in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        os.environ["SNMF_HSTEP_SR"] = hs
        os.environ["SNMF_WSTATS_SR"] = ws
        w, h, o = orig(V, q)
        d = np.abs(h - hr)
        t_bad = np.argsort(d.max(0))[-3:]
        print(f"it={it} hstep_sr={hs} wstats_sr={ws}: relW {fuzz_cases.rel(w, wr):.2e} relH {fuzz_cases.rel(h, hr):.2e}  worst frames {t_bad.tolist()} "
              f"max|dH| {d.max():.3e} at k={int(d.max(1).argmax())}  |H|max {np.abs(hr).max():.3e}", flush=True)
