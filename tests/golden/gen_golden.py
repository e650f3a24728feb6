#!/usr/bin/env python3
"""Regenerate tests/golden/*.json by running the UNMODIFIED reference.

The reference is compiled from /root/reference by `make -C oracle ref` into
oracle/_ref/ref_driver (git-ignored; the reference sources are never copied).
This script only works where /root/reference exists (the build container); the
JSON files it writes are committed and are what the tests read everywhere else.

Doubles are stored as C99 hexfloat strings (bit-exact); u64 as decimal strings.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def run(*args):
    out = subprocess.check_output([DRIVER, *map(str, args)], text=True)
    return json.loads(out)


def write(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as fh:
        json.dump(obj, fh, separators=(",", ":"))
        fh.write("\n")
    print(f"wrote {name}: {os.path.getsize(path)} bytes")


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(DRIVER):
        sys.exit("reference driver not built (is /root/reference present?)")

    # G1 — RNG streams
    write("rng.json", run("rng", 1024))

    # G2 — config C1 and relatives: full minimize() runs with default stops
    g2 = {
        "c1_random_pop40_x0_5_7": run("de", "random", 2, 40, 1000, 10e-4, 50, "5,7", 0),
        "random_pop50_x0_5_7": run("de", "random", 2, 50, 1000, 10e-4, 50, "5,7", 0),
        "example_best_pop50_x0_2_7": run("de", "best", 2, 50, 1000, 10e-4, 50, "2,7", 0),
        "readme_objective_pop40": run("de-readme", 40),
    }
    write("de_c1.json", g2)

    # G3 — per-evaluation traces, 5 generations, early stop disabled (eps=0)
    g3 = {}
    for strat in ("random", "best"):
        g3[f"{strat}_pop8_D4"] = run("de", strat, 4, 8, 5, 0, 1000, "2.5", 2)
        g3[f"{strat}_pop40_D2"] = run("de", strat, 2, 40, 5, 0, 1000, "5,7", 2)
        g3[f"{strat}_pop64_D16"] = run("de", strat, 16, 64, 5, 0, 1000, "4.096", 1)
        g3[f"{strat}_pop256_D128"] = run("de", strat, 128, 256, 5, 0, 1000, "4.096", 1)
    write("de_trace.json", g3)

    # G10 — the pass / fail matrix the reference's own test program prints (tests.cpp ->
    # test_functions.h:390-523: every solver with default arguments from x = (-0.5, -0.5),
    # "passed" = every coordinate within 0.05 of the known minimum; failures print the result)
    import re
    text = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_tests")], text=True)
    text = re.sub(r"\x1b\[[0-9;]*m", "", text)
    matrix, last = {}, None
    for line in text.splitlines():
        m = re.match(r"Solver (.+) on Problem (\S+) (passed|failed)\.", line)
        if m:
            last = matrix.setdefault(m.group(2), {}).setdefault(m.group(1), {})
            last["passed"] = m.group(3) == "passed"
            continue
        m = re.match(r"Result: (.*?)\s*\. Expected: (.*)", line)
        if m and last is not None:
            last["result"] = [float(v) for v in m.group(1).split()]
            last["expected"] = [float(v) for v in m.group(2).split()]
    write("reference_matrix.json", matrix)

    more = os.path.join(HERE, "gen_golden_more.py")
    if os.path.exists(more):
        import runpy
        runpy.run_path(more, init_globals={"run": run, "write": write})["main"]()
    # the distribution-level fixtures (de_stat.json, pso_stat.json, n4_stat.json: 128 seeded runs of
    # the reference per configuration) take three more minutes on 8 cores: --with-stat, or run
    # gen_golden_stat.py on its own; they regenerate byte-identically too
    stat = os.path.join(HERE, "gen_golden_stat.py")
    if "--with-stat" in sys.argv[1:]:
        subprocess.check_call([sys.executable, stat])
    else:
        print("(de_stat / pso_stat / n4_stat: python tests/golden/gen_golden_stat.py, or --with-stat)")


if __name__ == "__main__":
    main()
