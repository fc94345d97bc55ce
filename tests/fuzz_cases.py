"""Random shapes / modes through the product path against the fp64 oracle: the engine of scripts/fuzz_shapes.py (development runs,
any seed, minutes) and of tests/test_gpu_fuzz.py (fixed seeds inside the driver-run -m gpu suite).  Test infrastructure: imports the oracle.

focus = "r5": the envelopes of round 5's kernels -- k_iter_sf (F = 33..64, r = 65..128, KL, both factors updated) and
k_hstep_rp<., CUT> (>= 4 row tiles, r <= 64, more tiles than CUs) -- with their edge cases (partial last tile, F on both sides of
32n + 1, every sparsity form, cost on / off).  focus = "big": 33000..90000 frames (4..11 tiles per workgroup) on the main families:
every buffer of every pipeline wraps around several times (the race of profiles/r05_experiments.md section 10 only showed with three or
more tiles per workgroup).  focus = "pipe": every PIPELINED family (k_hstep_rp, its CUT / pair forms, k_hstep_rh modes 0 / 1 / 2,
k_hstep_sf, k_iter_sf, k_wstats with loader waves / teams, k_wstats_sf) at 3..6 tiles per workgroup with few iterations (cheap for the
oracle).  focus = "stop": early-stop cases on those families.  focus = "share": the shared tiles of the small-F family (round 6).  No focus: shapes drawn to land on the plan's geometry switches: tile counts
around multiples of the CU count (the split last round), F on both sides of 32n+1, r around the 32-column tiles and the LX / NK limits, all
divergences and update modes, shapes beyond the fused kernels' envelope now and then."""
import time

import numpy as np


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


class Fuzz:
    def __init__(self, seed, focus=""):
        self.rs = np.random.default_rng(seed)
        self.focus = focus
        self.fails, self.lines = [], []
        self.n_run = self.n_refused = self.n_borderline = self.n_unstable = 0

    # ---- shape draws ------------------------------------------------------------------------------------------------
    def draw_r5(self):
        rs = self.rs
        if rs.integers(0, 2) == 0:   # the fused small-F iteration
            F = int(rs.choice([64, 64, 40, 33, 63, int(rs.integers(33, 65))]))
            r = int(rs.choice([100, 128, 65, 96, 97, int(rs.integers(65, 129))]))
            T = int(rs.choice([int(rs.integers(1, 300)), int(rs.integers(300, 9000)), int(8192 + 32 * rs.integers(1, 700) + rs.integers(-31, 1)),
                               int(32 * 256 * rs.integers(1, 4) + rs.integers(-40, 40))]))
            mode = str(rs.choice(["full", "full", "full", "semi"]))
        else:                        # the contraction cut
            F = int(rs.choice([257, 513, 129, 128, 512, 385, 256, int(rs.integers(128, 514))]))
            r = int(rs.choice([20, 10, 30, 32, 33, 50, 64, 1, int(rs.integers(1, 65))]))
            T = int(8192 + 32 * rs.integers(1, 500) + rs.integers(-31, 1))
            mode = str(rs.choice(["full", "h", "h", "semi"]))
        sp = str(rs.choice(["scalar", "scalar", "vec", "mat", "zero"]))
        return F, T, r, 1.0, mode, sp

    def draw_big(self):
        """long solves: 4..11 tiles of 32 frames per workgroup on 256 CUs -- every buffer of every pipeline wraps around several times"""
        rs = self.rs
        F = int(rs.choice([257, 513, 129, 64, 64, 385, 512, 40, int(rs.integers(33, 65)), int(rs.integers(100, 514))]))
        r = int(rs.choice([20, 30, 32, 50, 64, 100, 100, 128, 200, 256, int(rs.integers(1, 257))]))
        T = int(rs.integers(33000, 90000))
        beta = float(rs.choice([1.0, 1.0, 1.0, 1.0, 2.0, 0.0, 0.5]))
        mode = str(rs.choice(["full", "full", "h", "w", "semi"]))
        sp = str(rs.choice(["scalar", "scalar", "vec", "zero"]))
        return F, T, r, beta, mode, sp

    def draw_pipe(self):
        """one family per draw, KL, 3..6 tiles per workgroup (8192 frames = one tile per workgroup on 256 CUs)"""
        rs = self.rs
        fam = int(rs.integers(0, 8))
        if fam == 0:    # k_hstep_rp + k_wstats<8, 4, 4, 2>: 5..8 row tiles, 129..256 components
            F, r = int(rs.choice([257, 256, 225, 161, 193, 255])), int(rs.choice([256, 200, 129, 160, 250, int(rs.integers(129, 257))]))
        elif fam == 1:  # the contraction cut, four ways / pairs
            F, r = int(rs.choice([513, 257, 385, 129, 512, 481])), int(rs.choice([20, 10, 30, 32, 33, 50, 64, int(rs.integers(1, 65))]))
        elif fam == 2:  # k_hstep_rh modes 0 / 1 / 2 + the eight-consumer statistics with leftover columns
            F, r = int(rs.choice([513, 512, 385, 449, 289])), int(rs.choice([100, 97, 200, 194, 140, 256, 128, int(rs.integers(65, 257))]))
        elif fam == 3:  # the small-F family, fused iteration
            F, r = int(rs.choice([64, 40, 33, 63])), int(rs.choice([100, 128, 65, 96, int(rs.integers(65, 129))]))
        elif fam == 4:  # the small-F family, two launches (r > 128 or F <= 32) and consumer teams
            F, r = int(rs.choice([64, 32, 64, 20])), int(rs.choice([200, 240, 130, 40, 256, int(rs.integers(1, 257))]))
        elif fam == 5:  # k_hstep_rp on few row tiles / few column tiles
            F, r = int(rs.choice([65, 97, 129, 128])), int(rs.choice([70, 96, 100, 40, 24]))
        elif fam == 6:  # the reference's shipped geometry, exactly
            F, r = 513, int(rs.choice([100, 200, 140, 50, 20]))
        else:
            F, r = int(rs.integers(33, 514)), int(rs.integers(1, 257))
        T = int(8192 * rs.integers(3, 7) + rs.integers(-4000, 4000))
        while F * T * r > 2.5e9:
            T = int(T * 0.8)
        mode = str(rs.choice(["full", "full", "h", "w", "semi"]))
        sp = str(rs.choice(["scalar", "scalar", "vec", "zero"]))
        return F, T, r, 1.0, mode, sp

    def draw_share(self):
        """Round 6's shared tiles of the small-F family (csrc/snmf_smallf.h): k_hstep_sf shares the tiles of a partial wave level that is
        the first on its SIMDs (tile counts 1025..1279 and 2049..2303 on 256 compute units) among four waves, k_wstats_sf a workgroup's
        single remainder tile (4 n + 1 tiles in a chunk) among its eight waves; H-only, W-only and (r > 128) full updates."""
        rs = self.rs
        F = int(rs.choice([64, 64, 40, 33, 63, 48, 32, 20]))
        mode = str(rs.choice(["h", "h", "w", "w", "full", "semi"]))
        if mode == "w":
            r = int(rs.choice([100, 128, 40, 64, 65, 96, int(rs.integers(33, 129))]))
        else:
            r = int(rs.choice([200, 240, 100, 140, 256, 70, 160, int(rs.integers(33, 257))]))
        tiles = int(rs.choice([1024, 2048])) + int(rs.integers(1, 256)) if mode != "w" else int(rs.choice([1024, 2048, 1280, 2304])) + int(rs.integers(1, 256))
        T = 32 * tiles - int(rs.integers(0, 32))
        sp = str(rs.choice(["scalar", "scalar", "vec", "zero", "mat"]))
        return F, T, r, 1.0, mode, sp

    def draw(self):
        rs = self.rs
        if self.focus == "r5":
            return self.draw_r5()
        if self.focus == "share":
            return self.draw_share()
        if self.focus == "big":
            return self.draw_big()
        if self.focus in ("pipe", "stop"):
            return self.draw_pipe()
        F = int(rs.choice([257, 513, 129, 65, 64, 128, 512, 385, 97, 64, 40, 32, int(rs.integers(8, 65)), int(rs.integers(8, 600)), int(rs.integers(8, 200))]))
        r = int(rs.choice([int(rs.integers(1, 40)), int(rs.integers(90, 132)), int(rs.integers(190, 260)), 100, 200, 256, 40,
                           int(rs.integers(257, 700)), int(rs.integers(1, 300))]))
        kind = rs.integers(0, 4)
        if kind == 0:
            T = int(rs.integers(1, 400))
        elif kind == 1:
            T = int(8192 * rs.integers(1, 4) + rs.integers(-3000, 3000))
        elif kind == 2:
            T = int(rs.integers(400, 9000))
        else:
            T = int(8192 + 32 * rs.integers(1, 140) + rs.integers(-31, 1))
        big = rs.integers(0, 12)  # now and then a shape beyond the fused kernels' envelope (csrc/snmf_generic.h)
        if big == 0:
            F, r, T = int(rs.integers(2500, 3200)), int(rs.integers(1, 80)), int(rs.integers(1, 700))
        elif big == 1:
            F, r, T = int(rs.choice([129, 200, 257, 64])), int(rs.integers(1030, 1300)), int(rs.integers(1, 5000))
        while F * T * r > 6e9 or ((F + r) > 2400 and big > 1):
            T = max(1, T // 2)
            if (F + r) > 2400:
                r = r // 2
        beta = float(rs.choice([1.0, 1.0, 1.0, 2.0, 0.0, 0.5, 1.5]))
        mode = str(rs.choice(["full", "full", "h", "w", "semi"]))
        sp = str(rs.choice(["scalar", "scalar", "vec", "mat", "zero"]))
        return F, T, r, beta, mode, sp

    # ---- one case ----------------------------------------------------------------------------------------------------
    def case(self, ci, log=print):
        from oracle.sparse_nmf_oracle import sparse_nmf as onmf
        from se_snmf_nat_amd import SnmfError, sparse_nmf
        rs, focus = self.rs, self.focus
        F, T, r, beta, mode, sp = self.draw()
        V = (rs.gamma(0.5, 1.0, (F, 12)) @ rs.gamma(0.3, 1.0, (12, T)) + 1e-3)
        dv = int(rs.integers(0, 8))  # data variants: spectrogram-like dynamic range, global scale, silent rows / frames
        if dv == 0:
            V = V ** 3 * 1e3
        elif dv == 1:
            V = V * float(rs.choice([1e-5, 1e5]))
        elif dv == 2:
            V[rs.random(F) < 0.1, :] = 0.0
            V[:, rs.random(T) < 0.05] = 0.0
        W0 = rs.random((F, r))
        H0 = rs.random((r, T))
        iters = int(rs.integers(2, 6)) if focus not in ("pipe", "share") else int(rs.integers(2, 4))
        # early stop (src/sparse_nmf.m:272-284): now and then a solve that may stop by itself -- the stop index must be the oracle's unless the
        # oracle's own decision was within 2 % of the threshold at some iteration (then the case is counted as borderline, not compared)
        eps = float(rs.choice([0, 0, 0, 1e-3, 3e-3, 1e-2])) if focus not in ("big", "pipe", "share") else 0.0
        if focus == "stop":
            eps = float(rs.choice([1e-3, 3e-3, 1e-2, 3e-2]))
        if eps > 0:
            iters = int(rs.integers(8, 40)) if focus != "stop" else int(rs.integers(8, 20))
        p = dict(cf={1.0: "kl", 2.0: "ed", 0.0: "is"}.get(beta, "x"), beta=beta, max_iter=iters, conv_eps=eps,
                 cost_check=1 if eps > 0 else int(rs.integers(0, 4) > 0), init_w=W0, init_h=H0)
        p["sparsity"] = {"scalar": float(rs.choice([0.1, 1.0, 5.0])), "zero": 0.0, "vec": rs.random(r) * 4,
                         "mat": rs.random((r, T)) * 3}[sp]
        if mode == "h":
            p["w_update_ind"] = np.zeros(r, bool)
        elif mode == "w":
            p["h_update_ind"] = np.zeros(r, bool)
        elif mode == "semi":
            p["w_update_ind"] = np.arange(r) >= r // 2
        tag = f"F={F} T={T} r={r} beta={beta} {mode} sp={sp} it={iters} eps={eps:g} cc={p['cost_check']} dv={dv}"
        try:
            w, h, o = sparse_nmf(V, p)
        except SnmfError as e:
            self.n_refused += 1
            log(f"{ci:3d} {tag}: REFUSED {str(e)[:90]}")
            return
        wr, hr, orf = onmf(V, p)
        self.n_run += 1
        # A solve on which the REFERENCE iteration itself is not contracting (its own cost goes UP by more than 1 % somewhere: the
        # multiplicative update is only monotone for beta in [1, 2], src/sparse_nmf.m:196-205 uses it for every beta) separates an
        # fp32 from an fp64 trajectory exponentially -- seed 605's case 9 (Itakura-Saito, W-only, 32 iterations, cost 6e8 -> 1.5e19 ->
        # 1.6e12): 2.5e-8 after one iteration, 3e-4 after 32.  Counted, not compared.
        if p["cost_check"] and len(orf["cost"]) > 1:
            c = np.asarray(orf["cost"], float)
            if np.any(c[1:] > 1.01 * c[:-1]) or not np.all(np.isfinite(c)):
                self.n_unstable += 1
                log(f"{ci:3d} {tag}: the oracle's own cost is not monotone (unstable iteration; skipped)")
                return
        if eps > 0:
            c = np.asarray(orf["cost"], float)
            rc = np.abs(np.diff(c)) / np.abs(c[:-1]) if len(c) > 1 else np.array([])
            if len(rc) and np.min(np.abs(rc - eps)) < 0.02 * eps:
                self.n_borderline += 1
                log(f"{ci:3d} {tag}: borderline stop decision in the oracle itself (skipped)")
                return
            if o["n_iter"] != orf["n_iter"]:
                log(f"{ci:3d} {tag}: n_iter {o['n_iter']} != oracle {orf['n_iter']}  <<< FAIL")
                self.fails.append(tag + " (stop index)")
                return
        n = min(len(o["cost"]), len(orf["cost"]))
        ec = float(np.max(np.abs(o["cost"][:n] - orf["cost"][:n]) / np.abs(orf["cost"][:n]))) if n and p["cost_check"] else 0.0
        ew, eh = rel(w, wr), rel(h, hr)
        bad = (not np.isfinite(w).all()) or (not np.isfinite(h).all()) or ew > 2e-4 or eh > 2e-4 or ec > 2e-5
        log(f"{ci:3d} {tag}: relW {ew:.1e} relH {eh:.1e} cost {ec:.1e}{'  <<< FAIL' if bad else ''}")
        if bad:
            self.fails.append(tag)

    def run(self, n_cases, budget_s, log=print):
        t_start = time.time()
        for ci in range(n_cases):
            if time.time() - t_start > budget_s:
                log(f"(time budget reached after {ci} cases)")
                break
            self.case(ci, log)
        return self.fails
