"""Distribution-level parity against the reference itself (SURVEY §0.1 "(b) statistical /
convergence-level at scale").

tests/golden/{de,pso}_stat.json hold K = 128 seeded runs of the UNMODIFIED reference per
configuration (tests/golden/gen_golden_stat.py). A sampler — the synchronous oracle on the CPU
(tests/test_stat_oracle.py) or the device engines through the C-ABI (tests/test_stat_gpu.py),
which agree bit for bit — produces K runs with distinct counter seeds; `compare` then tests each
statistic of the two samples:

  * two-sample Kolmogorov–Smirnov, p >= P_MIN, wherever the two algorithms are expected to be the
    same distribution (everything except the cases below);
  * where a KS test of 128 against 128 runs DOES resolve a difference — DE at pop 40, D 2, where a
    synchronous generation needs about 1.13x the reference's in-place generations to reach the
    std_err stop, and the D = 128 configurations with CR 0.1 (RATIO_BANDS below) — the measured
    median ratio is asserted inside a band around the figure DESIGN.md §3 records, so the
    difference is pinned rather than hidden.
"""
import json
import os

import numpy as np
from scipy.stats import ks_2samp

HERE = os.path.dirname(os.path.abspath(__file__))
P_MIN = 1e-3  # per statistic; about 60 statistics are tested, so a true match fails < 6 % of seeds sets
GOLDEN = 0x9E3779B97F4A7C15


def seed_of(k):
    """Counter seed of run k: distinct, and nothing to do with the reference's xorshift states."""
    return (12374563468 + GOLDEN * (k + 1)) & (2**64 - 1)


def load(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        g = json.load(fh)

    def hx(a):
        return np.array([float.fromhex(v) for v in a])

    for c in g["configs"].values():
        c["iters"] = np.array(c["iters"])
        c["f"] = hx(c["f"])
        c["best_after"] = {int(k): hx(v) for k, v in c["best_after"].items()}
        c["mean_after"] = {int(k): hx(v) for k, v in c["mean_after"].items()}
    return g


def x0_of(c):
    return (np.array([float(v) for v in c["x0"].split(",")]) if "," in c["x0"]
            else np.full(c["D"], float(c["x0"])))


class Sample:
    def __init__(self, gens):
        self.iters, self.f = [], []
        self.best_after = {g: [] for g in gens}
        self.mean_after = {g: [] for g in gens}

    def arrays(self):
        self.iters, self.f = np.array(self.iters), np.array(self.f)
        for d in (self.best_after, self.mean_after):
            for g in d:
                d[g] = np.array(d[g])
        return self


def run_marks(gens, max_iter, advance, snapshot, finish, sample):
    """One run: advance to each mark g (a stopped engine ignores further turns, so a run that
    stopped earlier reports its final state there, as the goldens do), then to the stop."""
    at = 0
    for g in gens:
        g = min(g, max_iter)
        if g > at:
            advance(g - at)
            at = g
        best, mean = snapshot()
        _append(sample, gens, best, mean)
    if max_iter + 1 > at:
        advance(max_iter + 1 - at)  # one extra turn: the head that fires the max_iter stop
    it, f = finish()
    sample.iters.append(it)
    sample.f.append(f)


def _append(sample, gens, best, mean):
    # marks are visited in order; fill the first mark that is still short
    n = len(sample.iters)
    for gg in gens:
        if len(sample.best_after[gg]) == n:
            sample.best_after[gg].append(best)
            sample.mean_after[gg].append(mean)
            return
    raise AssertionError("more snapshots than marks")


# Where 128-vs-128 runs resolve a real difference between the synchronous generation and the
# reference's in-place one (nlsolver.h:2449-2472: agent i's trial already sees the survivors of
# agents < i of the same generation): DE at pop 40, D 2, and the two D = 128 configurations with CR 0.1
# below. (config, statistic) -> (lo, hi) band on
# median(sample) / median(reference). Measured with the synchronous oracle = the device
# (DESIGN.md §3 "Distribution-level parity"): iterations-to-stop 49 vs 43 (1.14) for strategy
# random, 48 vs 43 (1.12) for best; population mean after 10 generations 1.12 / 1.40.
RATIO_BANDS = {
    ("random_pop40_D2", "iters"): (1.02, 1.30),
    ("best_pop40_D2", "iters"): (1.02, 1.30),
    ("random_pop40_D2", "mean_after_10"): (0.85, 1.70),
    ("best_pop40_D2", "mean_after_10"): (0.85, 1.70),
    # D = 128, CR 0.1 / F 0.5 (the D = 128 configuration in which trials are accepted): over 1024 x
    # 128 coordinates the population mean is so sharply determined (inter-quartile range 1.7 % of
    # its value) that 128 runs resolve a 1.2 % lag of the synchronous generation after 200
    # generations (16 291 vs 16 099); at 10 and 50 generations nothing is resolved (KS p 0.12, 0.06).
    ("random_pop1024_D128_CR0.1_F0.5_fixed200", "mean_after_200"): (0.995, 1.03),
}
# Strategy best with CR 0.1 is a premature-convergence regime: nine coordinates in ten of every
# trial are the best agent's, so the population collapses onto it within ten generations and
# stays there (mean = best from generation 10 to 200, in the reference and on the device alike).
# The reference collapses DURING the first generation — agent i's donors are already the near-copies
# of the best that agents < i accepted (nlsolver.h:2466-2471) — while a synchronous generation
# still draws all its donors from the diverse initial population: it freezes 16 % LOWER
# (19 333 vs 23 103, KS p 3e-32). Pinned as a ratio, every statistic of the configuration.
for _lab in ("f", "best_after_10", "mean_after_10", "best_after_50", "mean_after_50",
             "best_after_200", "mean_after_200"):
    RATIO_BANDS[("best_pop1024_D128_CR0.1_F0.5_fixed200", _lab)] = (0.75, 0.93)
# Not tested on their own: at pop 40 the std_err stop fires around generation 43 (reference) /
# 49 (synchronous), so "after 50 generations" is the final state for most reference runs and not
# yet for the synchronous ones — the iterations band above already states that difference.
SKIP = {("random_pop40_D2", "best_after", 50), ("best_pop40_D2", "best_after", 50),
        ("random_pop40_D2", "mean_after", 50), ("best_pop40_D2", "mean_after", 50)}


def compare(name, ref, smp, gens, report=None):
    """Assert every statistic; returns the list of (statistic, p or ratio) for the report."""
    out, bad = [], []

    def ks(label, a, b, key):
        if key in SKIP:
            return
        if (name, label) in RATIO_BANDS:
            lo, hi = RATIO_BANDS[(name, label)]
            r = float(np.median(a) / np.median(b))
            out.append((label, "ratio", r))
            if not lo <= r <= hi:
                bad.append(f"{label}: median ratio {r:.3f} outside [{lo}, {hi}]")
            return
        if np.array_equal(np.sort(a), np.sort(b)):
            p = 1.0
        else:
            p = float(ks_2samp(a, b).pvalue)
        out.append((label, "ks_p", p))
        if p < P_MIN:
            bad.append(f"{label}: KS p = {p:.3g} (medians {np.median(a):.6g} vs {np.median(b):.6g})")

    ks("iters", smp.iters, ref["iters"], (name, "iters"))
    ks("f", smp.f, ref["f"], (name, "f"))
    for g in gens:
        ks(f"best_after_{g}", smp.best_after[g], ref["best_after"][g], (name, "best_after", g))
        ks(f"mean_after_{g}", smp.mean_after[g], ref["mean_after"][g], (name, "mean_after", g))
    if report is not None:
        report[name] = {
            "median_iters": [float(np.median(smp.iters)), float(np.median(ref["iters"]))],
            "median_f": [float(np.median(smp.f)), float(np.median(ref["f"]))],
            "stats": [[a, b, c] for a, b, c in out],
        }
    assert not bad, f"{name}: " + "; ".join(bad)
    return out


# ---- the N4 rows (SANN, NelderMeadPSO): tests/golden/n4_stat.json ---------------------------------

N4_SEED = 12374563468  # one seed, run k = chain / instance k: the device keys its draws by (seed, chain)


def load_n4():
    with open(os.path.join(HERE, "golden", "n4_stat.json")) as fh:
        g = json.load(fh)
    for fam in ("sann", "nmpso"):
        for c in g[fam].values():
            c["f"] = np.array([float.fromhex(v) for v in c["f"]])
            c["iters"], c["fcalls"] = np.array(c["iters"]), np.array(c["fcalls"])
    return g


def significant(a, digits=12):
    """Values to `digits` significant digits: a run that never leaves its start point returns f(x0),
    which the reference sums in index order and the device as a lane tree — the same number to
    north_star's 1e-12, not the same bits. Without this the KS test reads 128 identical values
    against 128 identical values one ulp away as two different distributions (nmpso n = 32: every
    reference run returns 9.8645556679802162)."""
    a = np.asarray(a, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mag = np.where(a == 0, 1.0, 10.0 ** (digits - 1 - np.floor(np.log10(np.abs(a)))))
    return np.round(a * mag) / mag


def compare_n4(name, ref, f, iters, fcalls, report=None):
    out, bad = [], []
    for label, a, b in (("f", significant(f), significant(ref["f"])), ("iters", iters, ref["iters"]),
                        ("fcalls", fcalls, ref["fcalls"])):
        a, b = np.asarray(a), np.asarray(b)
        p = 1.0 if np.array_equal(np.sort(a), np.sort(b)) else float(ks_2samp(a, b).pvalue)
        out.append((label, p))
        if p < P_MIN:
            bad.append(f"{label}: KS p = {p:.3g} (medians {np.median(a):.6g} vs {np.median(b):.6g})")
    if report is not None:
        report[name] = {"median_f": [float(np.median(f)), float(np.median(ref["f"]))],
                        "stats": [[a, "ks_p", b] for a, b in out]}
    assert not bad, f"{name}: " + "; ".join(bad)
    return out
