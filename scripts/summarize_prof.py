#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by scripts/prof.sh on the GPU box) into the committed
summaries under profiles/: the rocprofv3 --kernel-trace --stats table, the PMC averages per kernel
and the per-launch HBM traffic used for bench.py's roofline.traffic.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from
SEPARATE --pmc passes; both are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of
wide (16 B/lane) coalesced streaming reads -- all bulk reads of these kernels are of that kind
(tile staging and W-fragment loads are f32x4 per lane) -- so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16 B/lane stores."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cmd = sys.argv[2] if len(sys.argv) > 2 else "bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(name):
    n = name.replace("void snmf::", "").replace("snmf::", "")
    return n.split("(")[0]


stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}\n")
    f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
    for r in rows:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("pmc1", "pmc2", "pmc3"):
    for fn in glob.glob(f"{src}/{p}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
traffic = {}
with open(f"profiles/{tag}_pmc.csv", "w") as f:
    f.write("# rocprofv3 --pmc <counters> (three separate passes: SQ/GRBM, FETCH_SIZE, WRITE_SIZE); averages per dispatch\n")
    cols = sorted({c for d in avg.values() for c in d})
    f.write("kernel," + ",".join(cols) + ",hbm_read_bytes(2*FETCH*1024),hbm_write_bytes(WRITE*1024)\n")
    for k, d in sorted(avg.items()):
        if not any(x in k for x in ("k_hstep", "k_wstats", "k_iter_sf", "k_reduce", "k_wapply", "k_wfin", "k_hsolve", "k_wadapt", "k_o")):
            continue
        rd = 2 * d.get("FETCH_SIZE", 0) * 1024
        wr = d.get("WRITE_SIZE", 0) * 1024
        traffic[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
        f.write(f"\"{k}\"," + ",".join(f"{d.get(c, float('nan')):.6g}" for c in cols) + f",{rd:.6g},{wr:.6g}\n")
calls = {short(r["Name"]): int(r["Calls"]) for r in rows}
for k in traffic:  # launches of each instantiation in the TRACE pass: what bench.pick_traffic_key selects by
    traffic[k]["calls"] = calls.get(k, 0)
traffic["_command"] = cmd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_digest: the sources this box's library was built from travel with the snapshot)
traffic["_csrc_digest"] = bench.csrc_digest()
if os.environ.get("SNMF_SOURCE_COMMIT"):
    traffic["_source_commit"] = os.environ["SNMF_SOURCE_COMMIT"]
json.dump(traffic, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
for fn in ("bench_trace.json",):
    if os.path.exists(f"{src}/{fn}"):
        open(f"profiles/{tag}_{fn}", "w").write(open(f"{src}/{fn}").read())
print(open(f"profiles/{tag}_kernel_stats.csv").read())
print(json.dumps(traffic, indent=1))
with open(f"profiles/{tag}_mfma_util.txt", "w") as fu:
    for k, d in avg.items():
        if ("k_hstep" in k or "k_wstats" in k or "k_iter_sf" in k) and "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            fu.write("%s MFMA pipe utilisation %.1f%% (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)\n"
                     % (k, 100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)))
for k, d in avg.items():
    if "k_hstep" in k or "k_wstats" in k or "k_iter_sf" in k:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            # MFMA busy cycles are summed over all 1024 SIMDs; GUI_ACTIVE is summed over 8 XCDs
            util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)
            print(k, "MFMA pipe utilisation %.1f%%" % (100 * util))
