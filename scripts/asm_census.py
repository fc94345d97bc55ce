#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing: per basic block the number of MFMA / other VALU / SALU /
LDS / VMEM / waitcnt instructions, loops (blocks that branch backwards) marked.  Development aid for the role pipelines:
under v_mfma_f32_32x32x2_f32 a SIMD is a sequential machine (profiles/r02_experiments.md), so what a tile costs beyond its
MFMAs is the issue cycles of everything else -- this says where that everything else sits.

    hipcc ... --cuda-device-only -S csrc/snmf_tu_hstep_rp.hip -o rp.s ; python scripts/asm_census.py rp.s k_hstep_rpILb1 [min_mfma]
"""
import re
import sys

fn, pat = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(fn).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
# also run to the last s_endpgm before .section
blocks, cur = [], {"label": "entry", "ins": [], "line": start}
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "ins": [], "line": i}
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur["ins"].append(s.split(";")[0].strip())
blocks.append(cur)
idx = {b["label"]: k for k, b in enumerate(blocks)}


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "br"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


tot = {}
for k, b in enumerate(blocks):
    c = {}
    back = False
    for ins in b["ins"]:
        op = ins.split()[0]
        t = cls(op)
        c[t] = c.get(t, 0) + 1
        tot[t] = tot.get(t, 0) + 1
        if t == "br":
            tgt = ins.split()[-1]
            if tgt in idx and idx[tgt] <= k:
                back = True
    b["c"], b["back"] = c, back
    if c.get("mfma", 0) >= min_mfma:
        print(f"{b['label']:>12} @{b['line']:6d} {'LOOP' if back else '    '} " + " ".join(f"{t}={c[t]}" for t in
              ("mfma", "valu", "acc", "salu", "lds", "vmem", "wait", "nop", "smem", "br", "other") if c.get(t)))
print("TOTAL", tot)
