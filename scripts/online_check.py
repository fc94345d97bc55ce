"""Dev check: device online separation vs the oracle on the fixture audio (prints the first mismatches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.online_oracle import default_params, ntf_sep_event_rt as oracle_rt
from se_snmf_nat_amd.online import OnlineSeparator, default_settings, ntf_sep_event_rt

d = np.load(os.path.join(ROOT, "tests/golden/ref_data.npz"))
B = d["B"].astype(np.float64)
s = np.load(os.path.join(ROOT, "tests/golden/frontend_audio.npz"))["samples"]
nfr = int(os.environ.get("NFR", "124"))
s = s[: max(0, nfr - 4) * 160]
p = default_params()
for kv in os.environ.get("OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        p[k] = type(p[k])(v) if not isinstance(p[k], str) else v
rs = np.random.RandomState(1)
H0 = rs.random_sample(200); Ad0 = rs.random_sample((50, 100))
t = time.time()
o16, of, Bd, tr = oracle_rt(s, B[:, :100], B[:, 100:], p, H0, Ad0, return_trace=True)
t_or = time.time() - t
ps = default_settings(); ps.update({k: p[k] for k in p if k in ps})
sep = OnlineSeparator(B[:, :100], B[:, 100:], ps, H0=H0, Ad_blk0=Ad0)
t = time.time()
out = sep.process(s, flush=True)
t_dev = time.time() - t
tg = sep.trace()
print("frames", len(tr), len(tg), "oracle %.2fs device %.3fs" % (t_or, t_dev))
bad = 0
for i, (a, b) in enumerate(zip(tr, tg)):
    same = (a["n_iter"] == b["n_iter"] and int(a["trig"]) == b["trig"] and a["n_up"] == b["n_up"] and int(a["solved"]) == b["solved"]
            and a["adapt_iters"] == b["adapt_iters"])
    if not same or i < 3:
        print(i + 1, "oracle", a["n_iter"], int(a["trig"]), a["n_up"], int(a["solved"]), a["adapt_iters"], "%.4g %.4g" % (a["beta"], a["A_x_mag"]),
              "| dev", b["n_iter"], b["trig"], b["n_up"], b["solved"], b["adapt_iters"], "%.4g %.4g" % (b["beta"], b["A_x_mag"]))
        bad += not same
        if bad > 8:
            break
xf = out["x_tilde_f"].astype(np.float64)
print("out len", len(of), len(xf))
n = min(len(of), len(xf))
err = np.abs(of[:n] - xf[:n])
print("max abs err %.4g  rel fro %.3g  max|x| %.1f  int16 max diff %d" % (err.max(), np.linalg.norm(of[:n] - xf[:n]) / np.linalg.norm(of[:n]),
      np.abs(of).max(), np.abs(o16[:n].astype(int) - out["x_tilde"][:n].astype(int)).max()))
hop = 160
for j in range(0, n // hop, max(1, n // hop // 12)):
    print(" hop", j, "err %.3g" % err[j * hop:(j + 1) * hop].max(), "ref %.1f" % np.abs(of[j * hop:(j + 1) * hop]).max())
print("basis rel err", np.linalg.norm(sep.basis() - Bd) / np.linalg.norm(Bd))
