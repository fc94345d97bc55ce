"""bench.py's record of the dominant kernel's HBM traffic: the entry quoted from the committed
rocprofv3 --pmc summaries must be the kernel the line names, at the size the timed region runs (CPU test; no GPU)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json"))))
def test_traffic_entry_is_the_named_kernel_at_full_size(fn):
    tj = json.load(open(fn))
    F, T, r = bench.F_, bench.T_, bench.R_
    for fam, kname in (("hstep", "k_hstep_rp"), ("wstats", "k_wstats")):
        key = bench.pick_traffic_key(tj, kname)
        if key is None:  # round 1 ran another H-step geometry
            continue
        assert key.startswith(kname + "<")
        # never a warm-up / drop-in launch of a few thousand frames: at least the algorithmic bytes of the full-size launch
        assert tj[key]["total_bytes"] >= bench.algorithmic_bytes(fam, F, T, r), (key, tj[key])
        assert tj[key]["total_bytes"] < 3 * bench.algorithmic_bytes(fam, F, T, r)


def test_a_family_prefix_is_not_a_match():
    tj = {"k_hstep<8, 1, 4, 1, true, true, false, 32>": {"total_bytes": 17e6},
          "k_hstep_rp<true>": {"total_bytes": 340e6}, "k_hstep_rp<false>": {"total_bytes": 341e6}, "_command": "x"}
    assert bench.pick_traffic_key(tj, "k_hstep_rp") == "k_hstep_rp<true>"
    assert bench.pick_traffic_key(tj, "k_hstep") == "k_hstep<8, 1, 4, 1, true, true, false, 32>"
    assert bench.pick_traffic_key(tj, "k_wstats") is None
    for k in tj:
        if not k.startswith("_"):
            tj[k]["calls"] = 3
    tj["k_hstep_rp<false>"]["calls"] = 400  # a run without cost_check: the variant launched most often wins
    assert bench.pick_traffic_key(tj, "k_hstep_rp") == "k_hstep_rp<false>"


def test_committed_traffic_of_the_newest_round():
    tr, src, frac = bench.committed_traffic("k_hstep_rp", 4.0 * bench.F_ * bench.T_ * bench.R_,
                                            bench.algorithmic_bytes("hstep", bench.F_, bench.T_, bench.R_))
    assert src["kernel"].startswith("k_hstep_rp<") and tr >= bench.algorithmic_bytes("hstep", bench.F_, bench.T_, bench.R_)
    assert 0.5 < frac < 1.0 and src["rocprof_calls"] > 100
