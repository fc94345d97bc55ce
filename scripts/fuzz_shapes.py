"""Development helper: random shapes / modes through the product path against the fp64 oracle (GPU box).
   python scripts/fuzz_shapes.py [seed] [n_cases] [time budget s] [focus]  -> one line per case, FAIL lines at the end.
The engine (shape draws, data variants, tolerances) is tests/fuzz_cases.py -- the same code tests/test_gpu_fuzz.py runs with fixed seeds
inside the driver's -m gpu suite; focus: "", "r5", "big", "pipe", "stop" (described there)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import Fuzz  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 420.0
focus = sys.argv[4] if len(sys.argv) > 4 else ""
fz = Fuzz(seed, focus)
fails = fz.run(n_cases, budget, log=lambda s: print(s, flush=True))
print(f"{fz.n_run} cases compared ({fz.n_refused} refused, {fz.n_borderline} borderline stop decisions, {fz.n_unstable} unstable in the oracle itself), {len(fails)} failures")
for f in fails:
    print("FAIL", f)
sys.exit(1 if fails else 0)
