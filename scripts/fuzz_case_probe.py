"""Re-run ONE fuzz case (seed, focus, index) at several iteration counts: how the distance to the oracle grows.
   python scripts/fuzz_case_probe.py 605 "" 9"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fuzz_cases  # noqa: E402
from oracle.sparse_nmf_oracle import sparse_nmf as onmf  # noqa: E402
from se_snmf_nat_amd import sparse_nmf  # noqa: E402

seed, focus, idx = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
captured = {}
real = fuzz_cases.Fuzz.case


class Grab(fuzz_cases.Fuzz):
    pass


fz = fuzz_cases.Fuzz(seed, focus)
# replay the generator up to the case: monkey-patch sparse_nmf / oracle calls away for the earlier cases
import se_snmf_nat_amd  # noqa: E402
import oracle.sparse_nmf_oracle as om  # noqa: E402

calls = []
orig_s, orig_o = se_snmf_nat_amd.sparse_nmf, om.sparse_nmf


def fake_s(V, p, **kw):
    calls.append((V, p))
    raise se_snmf_nat_amd.SnmfError(0, "skipped")


se_snmf_nat_amd.sparse_nmf = fake_s
for ci in range(idx + 1):
    fz.case(ci, log=lambda s: None)
se_snmf_nat_amd.sparse_nmf = orig_s
V, p = calls[idx]
print({k: (v if np.isscalar(v) or isinstance(v, str) else getattr(v, "shape", v)) for k, v in p.items()}, V.shape, V.min(), V.max())
for it in (1, 2, 4, 8, 16, 32):
    q = dict(p, max_iter=it, conv_eps=0.0)
    w, h, o = sparse_nmf(V, q)
    wr, hr, orf = onmf(V, q)
    n = min(len(o["cost"]), len(orf["cost"]))
    ec = float(np.max(np.abs(o["cost"][:n] - orf["cost"][:n]) / np.abs(orf["cost"][:n]))) if n else 0.0
    print(it, "relW", fuzz_cases.rel(w, wr), "relH", fuzz_cases.rel(h, hr), "cost", ec, "cost_last", orf["cost"][-1] if n else None, flush=True)
