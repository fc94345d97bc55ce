"""The shape fuzzer inside the driver-run suite (round 5's only device-side bug -- a DMA refill racing another loader wave's copy-out,
errors of 1e-3 .. 1e-1, profiles/r05_experiments.md section 10 -- was found by scripts/fuzz_shapes.py while every unit test was green:
the unit-test shapes had at most two tiles per workgroup).  Fixed seeds, the focused envelopes of tests/fuzz_cases.py, every case
through the C ABI against the fp64 oracle (W, H at 2e-4, costs at 2e-5, stop index exact unless the oracle's own decision is borderline;
src/sparse_nmf.m:186-286).  A time cap per envelope keeps the module at about 150 s; the number of cases that ran is asserted so that a
slow box cannot turn the test into a no-op."""
import pytest

from fuzz_cases import Fuzz

pytestmark = pytest.mark.gpu

# (focus, seed, cases, time cap in s, fewest cases that must have been compared)
ENVELOPES = [("pipe", 601, 40, 55.0, 8), ("r5", 602, 40, 30.0, 8), ("big", 603, 12, 35.0, 2), ("stop", 604, 20, 25.0, 3), ("", 605, 40, 25.0, 8),
             ("share", 606, 14, 30.0, 4)]


@pytest.mark.parametrize("focus,seed,n_cases,cap,least", ENVELOPES, ids=[e[0] or "general" for e in ENVELOPES])
def test_random_shapes_against_the_oracle(gpu_ctx, focus, seed, n_cases, cap, least):
    fz = Fuzz(seed, focus)
    lines = []
    fails = fz.run(n_cases, cap, log=lines.append)
    assert not fails, "\n".join(ln for ln in lines if "FAIL" in ln)
    assert fz.n_run >= least, (fz.n_run, fz.n_refused, lines[-3:])
