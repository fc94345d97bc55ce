#!/usr/bin/env python3
"""One reference shape for a few iterations (scripts/reach.sh runs it under rocprofv3 --kernel-trace --stats, one invocation per
shape): which kernel instantiations does a setting of the reference actually launch?  -> profiles/r06_reachable.json"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from se_snmf_nat_amd import Context, Plan

# name: (F, r, mode, beta, sparsity form)   T = 72000 frames unless given (12 min of audio at the shipped settings)
SHAPES = {
    # settings/initial_setting_SNMF_NAT.m:48-49  R_x = R_d = 100, F = 513 (40 ms window), Mel 64 bands
    "nat_train_r100": (513, 100, "full", 1.0), "nat_dnmf_h_r200": (513, 200, "h", 1.0), "nat_dnmf_w_r100": (513, 100, "w", 1.0),
    "nat_mel_train_r100": (64, 100, "full", 1.0), "nat_mel_dnmf_h_r200": (64, 200, "h", 1.0), "nat_mel_dnmf_w_r100": (64, 100, "w", 1.0),
    # settings/bak_IS16_results/initial_setting_SNMF_Techwin_201603_RT.m:47-48  R_x = 20, R_d = 10
    "techwin_train_r20": (513, 20, "full", 1.0), "techwin_train_r10": (513, 10, "full", 1.0), "techwin_dnmf_h_r30": (513, 30, "h", 1.0),
    "techwin_dnmf_w_r20": (513, 20, "w", 1.0), "techwin_dnmf_w_r10": (513, 10, "w", 1.0),
    # settings/bak_IS16_results/initial_setting_IMCRA.m:47-48  R = 50
    "imcra_train_r50": (513, 50, "full", 1.0), "imcra_dnmf_h_r100": (513, 100, "h", 1.0), "imcra_dnmf_w_r50": (513, 50, "w", 1.0),
    # R_x = 140, R_d = 100
    "r140_train": (513, 140, "full", 1.0), "r140_dnmf_h_r240": (513, 240, "h", 1.0), "r140_dnmf_w_r140": (513, 140, "w", 1.0),
    "r140_mel_dnmf_h_r240": (64, 240, "h", 1.0), "r140_mel_train": (64, 140, "full", 1.0),
    # settings/bak_IS16_results/initial_setting_Exemplar.m:47-48  R_x = R_d = 500 (exemplars; the DNMF loop on them)
    "exemplar_dnmf_h_r1000": (513, 1000, "h", 1.0), "exemplar_dnmf_w_r500": (513, 500, "w", 1.0),
    # BASELINE configs
    "C1": (257, 40, "full", 1.0, 2000), "C2": (257, 256, "full", 1.0, 100000), "C5": (513, 512, "full", 2.0, 100000),
}

name = sys.argv[1]
c = SHAPES[name]
F, r, mode, beta = c[:4]
T = c[4] if len(c) > 4 else 72000
rs = np.random.default_rng(1)
V = (rs.gamma(0.5, 1.0, (F, 8)).astype(np.float32) @ rs.gamma(0.3, 1.0, (8, T)).astype(np.float32) + 1e-3)
kw = {}
if mode == "h": kw["w_update_ind"] = np.zeros(r, bool)
if mode == "w": kw["h_update_ind"] = np.zeros(r, bool)
ctx = Context(0)
pl = Plan(ctx, F, T, r, beta=beta, max_iter=4, conv_eps=0.0, cost_check=True, sparsity=5.0, **kw)
pl.set_v(V); pl.set_w(rs.random((F, r))); pl.set_h(rs.random((r, T)).astype(np.float32)); pl.init(); pl.run()
print(name, pl.describe())
pl.close()
