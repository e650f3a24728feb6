#!/usr/bin/env python3
"""Regenerate tests/golden/de_stat.json, pso_stat.json and n4_stat.json (SANN, NelderMeadPSO): K seeded runs of the UNMODIFIED
reference per configuration (oracle/_ref/ref_driver de-stat / pso-stat, which reseed the
reference's xorshift through its own set_state, nlsolver.h:1367).

Per run: iterations-to-stop, function calls, final f, and after g generations the best f and the
population's mean f (every member's lowest value so far).
Only works where /root/reference exists; the JSON written here is committed (data only).
Runtime: about 3 minutes on 8 cores.
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

K = 128
GENS = (10, 50, 200)

# (name, D, n, max_iter, eps, no_change, x0): the first three run to the reference's DEFAULT stop
# rule (DE ctor defaults nlsolver.h:2390-2394: eps 10e-4, max_iter 1000, no_change 50), the last
# runs a fixed number of generations with the stop tests disabled.
# With the ctor defaults CR 0.9 / F 0.8 no trial is ever accepted at D = 128 within 200 generations
# (reference and device alike: the population mean never moves), so a fifth configuration with
# CR 0.1 / F 0.5 — where the mean falls 5x in 200 generations — gives the D = 128 comparison
# something to resolve.
DE_CONFIGS = [
    ("pop40_D2", 2, 40, 1000, 10e-4, 50, "5,7", 0.9, 0.8),
    ("pop256_D16", 16, 256, 1000, 10e-4, 50, "4.096", 0.9, 0.8),
    ("pop1024_D32", 32, 1024, 1000, 10e-4, 50, "4.096", 0.9, 0.8),
    ("pop4096_D128_fixed200", 128, 4096, 200, 0.0, 1000, "4.096", 0.9, 0.8),
    ("pop1024_D128_CR0.1_F0.5_fixed200", 128, 1024, 200, 0.0, 1000, "4.096", 0.1, 0.5),
]
# PSO ctor defaults nlsolver.h:2522-2526 (max_iter 5000, no_change 50, eps 10e-4); x0 = 2.048 so
# the unbounded overload derives bounds +-2.048 (2553-2560).
PSO_CONFIGS = [
    ("n40_D2", 2, 40, 5000, 10e-4, 50, "2.048"),
    ("n256_D16", 16, 256, 5000, 10e-4, 50, "2.048"),
    ("n1024_D32", 32, 1024, 5000, 10e-4, 50, "2.048"),
    ("n4096_D128_fixed200", 128, 4096, 200, 0.0, 1000, "2.048"),
]


# SANN (ctor defaults nlsolver.h:2759-2761: max_iter 5000, temperature_iter 10, temperature_max 10)
# and NelderMeadPSO (3563-3569: eps 1e-6, max_iter 1000, no_change 20) on Rosenbrock from
# x_i = 0.5 + 0.01 i: (name, n, args of the driver's *-stat command)
SANN_CONFIGS = [
    ("n2", 2, (5000, 10, 10.0)),
    ("n16", 16, (5000, 10, 10.0)),
    ("n128_2000iters", 128, (2000, 10, 10.0)),
]
NMPSO_CONFIGS = [
    ("n2", 2, (1000, 1e-6, 20)),
    ("n8", 8, (1000, 1e-6, 20)),
    ("n32", 32, (1000, 1e-6, 20)),
]


def chunked_n4(cmd, n, args, chunks):
    def one(c):
        k0, k1 = c
        out = subprocess.check_output([DRIVER, cmd, "0", str(n), *map(repr, args), "0.5", "0.01",
                                       str(k0), str(k1)], text=True)
        return json.loads(out)["runs"]
    with ThreadPoolExecutor(max_workers=8) as ex:
        parts = list(ex.map(one, chunks))
    runs = [r for p in parts for r in p]
    return {"k": [r["k"] for r in runs], "iters": [r["iters"] for r in runs],
            "fcalls": [r["fcalls"] for r in runs], "f": [r["f"] for r in runs]}


def chunked(cmd, kind, D, n, max_iter, eps, no_change, x0, chunks, extra=()):
    def one(c):
        k0, k1 = c
        out = subprocess.check_output(
            [DRIVER, cmd, kind, str(D), str(n), str(max_iter), repr(eps), str(no_change), x0,
             str(k0), str(k1), ",".join(map(str, GENS)), *map(repr, extra)], text=True)
        return json.loads(out)["runs"]
    with ThreadPoolExecutor(max_workers=8) as ex:
        parts = list(ex.map(one, chunks))
    return [r for p in parts for r in p]


def pack(runs):
    """Column form: smaller file, same data."""
    return {
        "k": [r["k"] for r in runs],
        "iters": [r["iters"] for r in runs],
        "fcalls": [r["fcalls"] for r in runs],
        "f": [r["f"] for r in runs],
        "best_after": {str(g): [r["best_after"][i] for r in runs] for i, g in enumerate(GENS)},
        "mean_after": {str(g): [r["mean_after"][i] for r in runs] for i, g in enumerate(GENS)},
    }


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(DRIVER):
        sys.exit("reference driver not built (is /root/reference present?)")
    chunks = [(k, min(k + 8, K)) for k in range(0, K, 8)]
    de = {"K": K, "gens": list(GENS), "seed_rule": "splitmix(12374563468 + k) -> y, z -> set_state",
          "configs": {}}
    for name, D, n, mi, eps, nc, x0, CR, F in DE_CONFIGS:
        for strat in ("random", "best"):
            runs = chunked("de-stat", strat, D, n, mi, eps, nc, x0, chunks, (CR, F))
            de["configs"][f"{strat}_{name}"] = dict(
                strategy=strat, D=D, pop=n, max_iter=mi, eps=eps, no_change=nc, x0=x0, CR=CR, F=F,
                **pack(runs))
            print("de", strat, name, "done", flush=True)
    with open(os.path.join(HERE, "de_stat.json"), "w") as fh:
        json.dump(de, fh, separators=(",", ":"))
        fh.write("\n")
    pso = {"K": K, "gens": list(GENS), "seed_rule": de["seed_rule"], "configs": {}}
    for name, D, n, mi, eps, nc, x0 in PSO_CONFIGS:
        runs = chunked("pso-stat", "accelerated", D, n, mi, eps, nc, x0, chunks)
        pso["configs"][f"accelerated_{name}"] = dict(
            type="accelerated", D=D, particles=n, max_iter=mi, eps=eps, no_change=nc, x0=x0,
            **pack(runs))
        print("pso", name, "done", flush=True)
    with open(os.path.join(HERE, "pso_stat.json"), "w") as fh:
        json.dump(pso, fh, separators=(",", ":"))
        fh.write("\n")
    n4 = {"K": K, "seed_rule": de["seed_rule"], "x0": "0.5 + 0.01 i", "sann": {}, "nmpso": {}}
    for name, n, args in SANN_CONFIGS:
        n4["sann"][name] = dict(n=n, max_iter=args[0], temperature_iter=args[1], temperature_max=args[2],
                                **chunked_n4("sann-stat", n, args, chunks))
        print("sann", name, "done", flush=True)
    for name, n, args in NMPSO_CONFIGS:
        n4["nmpso"][name] = dict(n=n, max_iter=args[0], eps=args[1], no_change=args[2],
                                 **chunked_n4("nmpso-stat", n, args, chunks))
        print("nmpso", name, "done", flush=True)
    with open(os.path.join(HERE, "n4_stat.json"), "w") as fh:
        json.dump(n4, fh, separators=(",", ":"))
        fh.write("\n")


if __name__ == "__main__":
    main()
