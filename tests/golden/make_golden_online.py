"""Online-separation fixture (BASELINE config 3): the oracle's run of the driver loop
(src/NTF_sep_event_RT.m + src/bnmf_sep_event_RT_IS16.m, shipped settings) over the committed 1.2 s
audio fixture (tests/golden/frontend_audio.npz, cut from the reference's shipped wav) with the shipped
dictionaries (tests/golden/ref_data.npz: B = [B_DFT_x(:,1:100), B_DFT_d(:,1:100)]).
PARITY UNPINNED w.r.t. MATLAB (oracle/online_oracle.py).  H0 / Ad_blk0 = RandomState(1) draws, in that order.
Run from the repo root: python tests/golden/make_golden_online.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.online_oracle import default_params, ntf_sep_event_rt  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
B = np.load(os.path.join(G, "ref_data.npz"))["B"].astype(np.float64)
s = np.load(os.path.join(G, "frontend_audio.npz"))["samples"]
p = default_params()
rs = np.random.RandomState(1)
H0 = rs.random_sample(200)
Ad0 = rs.random_sample((50, 100))
o16, of, Bd, tr = ntf_sep_event_rt(s, B[:, :100], B[:, 100:], p, H0, Ad0, return_trace=True)
np.savez_compressed(
    os.path.join(G, "online_is16_124frames.npz"), x_tilde_i16=o16, x_tilde_f=of.astype(np.float32),
    B_DFT_d_sub=Bd[::4].astype(np.float32), B_DFT_d_fro=np.linalg.norm(Bd),
    n_iter=np.array([t["n_iter"] for t in tr], np.int32), trig=np.array([t["trig"] for t in tr], np.int8),
    n_up=np.array([t["n_up"] for t in tr], np.int32), adapt_iters=np.array([t["adapt_iters"] for t in tr], np.int32),
    beta=np.array([t["beta"] for t in tr]), A_x_mag=np.array([t["A_x_mag"] for t in tr]),
    Q_control=np.array([t["Q_control"] for t in tr]))
print(len(tr), "frames", o16.shape, os.path.getsize(os.path.join(G, "online_is16_124frames.npz")), "bytes")
