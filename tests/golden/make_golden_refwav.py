"""Reference-held pin of the online path: the only outputs of the reference's OWN code that the
reference repository holds are the two processed recordings next to their inputs,

    wav/M03_423C0213_STR.CH6.wav  ->  wav/M03_423C0213_STR.CH6_out_v3.9_18.wav   (filewise_run_IS16.m:6-10)
    wav/LM_in.wav                 ->  wav/LM_in_out_v3.9_18.wav                   (filewise_run_IS16.m:7, alternate fname)

written by src/NTF_sep_event_RT.m:54-139 under MATLAB with the shipped R_100 dictionaries.  This script
stores the int16 SAMPLES of the four files (data, not source) as tests/golden/refwav_pairs.npz so that the tests can
compare the oracle (CPU) and the HIP online path (GPU) with what MATLAB wrote, on the GPU box where
/root/reference does not exist.  The reader follows the reference's own: skip 22 int16 words of header, the rest
is PCM (src/NTF_sep_event_RT.m:54-58).

Run from the repo root (needs /root/reference):  python tests/golden/make_golden_refwav.py
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("SNMF_REFERENCE", "/root/reference")
PAIRS = {"m03": ("M03_423C0213_STR.CH6.wav", "M03_423C0213_STR.CH6_out_v3.9_18.wav"),
         "lm": ("LM_in.wav", "LM_in_out_v3.9_18.wav")}


def read_pcm(path):
    raw = np.fromfile(path, dtype="<i2")
    return raw[22:].copy()  # header = 22 int16 words, src/NTF_sep_event_RT.m:56


if __name__ == "__main__":
    out = {}
    for key, (fi, fo) in PAIRS.items():
        out[key + "_in"] = read_pcm(os.path.join(REF, "wav", fi))
        out[key + "_out"] = read_pcm(os.path.join(REF, "wav", fo))
        print(key, out[key + "_in"].shape, out[key + "_out"].shape)
    dst = os.path.join(ROOT, "tests", "golden", "refwav_pairs.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes")
