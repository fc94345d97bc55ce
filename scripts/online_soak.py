"""Dev check: long-stream behaviour of the device online loop vs the oracle (decision agreement, output error)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.online_oracle import default_params, ntf_sep_event_rt as oracle_rt
from se_snmf_nat_amd.online import OnlineSeparator, default_settings
B = np.load(os.path.join(ROOT, "tests/golden/ref_data.npz"))["B"].astype(np.float64)
s = np.load(os.path.join(ROOT, "tests/golden/frontend_audio.npz"))["samples"]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
s = np.tile(s, int(np.ceil(secs * 16000 / len(s))))[:int(secs * 16000)]
p = default_params()
rs = np.random.RandomState(1)
H0, Ad0 = rs.random_sample(200), rs.random_sample((50, 100))
t = time.time(); o16, of, Bd, tr = oracle_rt(s, B[:, :100], B[:, 100:], p, H0, Ad0, return_trace=True); t_or = time.time() - t
sep = OnlineSeparator(B[:, :100], B[:, 100:], default_settings(), H0=H0, Ad_blk0=Ad0)
t = time.time(); out = sep.process(s, flush=True); t_dev = time.time() - t
tg = sep.trace()
same = [a["n_iter"] == b["n_iter"] and a["adapt_iters"] == b["adapt_iters"] and a["n_up"] == b["n_up"] for a, b in zip(tr, tg)]
first = next((i for i, x in enumerate(same) if not x), None)
xf = out["x_tilde_f"].astype(np.float64)
hop = 160
errs = np.array([np.linalg.norm(xf[j*hop:(j+1)*hop] - of[j*hop:(j+1)*hop]) / max(np.linalg.norm(of[j*hop:(j+1)*hop]), 1.0) for j in range(len(of)//hop)])
print("frames", len(tr), "oracle %.1fs device %.2fs" % (t_or, t_dev), "first decision mismatch at frame", first, "agreeing frames %.1f%%" % (100*np.mean(same)))
print("overall rel err %.3g; per-hop rel err quantiles 50/90/99/max: %.2g %.2g %.2g %.2g" % (np.linalg.norm(xf-of)/np.linalg.norm(of), *np.quantile(errs, [0.5, 0.9, 0.99, 1.0])))
print("finite", np.isfinite(xf).all(), "int16 max diff", np.abs(out["x_tilde"].astype(int) - o16.astype(int)).max(), "basis rel", np.linalg.norm(sep.basis()-Bd)/np.linalg.norm(Bd))
for a0 in range(0, min(len(errs), 600), 50):
    print("hops %4d-%4d: median rel err %.2g max %.2g" % (a0, a0 + 49, np.median(errs[a0:a0+50]), errs[a0:a0+50].max()))
if first is not None:
    a, b = tr[first], tg[first]
    print("mismatch frame", first + 1, "oracle n_iter/adapt/n_up", a["n_iter"], a["adapt_iters"], a["n_up"], "device", b["n_iter"], b["adapt_iters"], b["n_up"])
