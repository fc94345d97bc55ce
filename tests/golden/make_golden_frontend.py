"""Front-end fixture: 1.2 s of int16 audio from the reference's shipped wav (a DATA file of the
reference: wav/M03_423C0213_STR.CH6.wav, samples 8000..27199) and the oracle's DFT / Mel features
of it for the shipped settings.  PARITY UNPINNED w.r.t. MATLAB (oracle/frontend_oracle.py).
Run from the repo root in the build container: python tests/golden/make_golden_frontend.py"""
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.frontend_oracle import default_params, dft_features, mel_features  # noqa: E402

with wave.open("/root/reference/wav/M03_423C0213_STR.CH6.wav") as wf:
    x = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
s = x[8000:27200].copy()
p = default_params()
V = dft_features(s.astype(np.float64), p)
M = mel_features(V, p)
p1 = dict(p, Splice=1, preemph=0.92, pow=1)
V1 = dft_features(s.astype(np.float64), p1)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frontend_audio.npz"), samples=s,
                    V_col_max=V.max(0), V_sub=V[::8, ::4], mel_sub=M[::4, ::4], V1_sub=V1[::16, ::4],
                    n_frames=V.shape[1])
print(V.shape, M.shape, V1.shape)
