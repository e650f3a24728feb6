#!/usr/bin/env python3
"""Where the kernels' scratch (register spill) accesses sit: for every kernel of the library that
has a non-zero ScratchSize, its VGPR count, scratch bytes per lane, and how many of its
scratch_load / scratch_store instructions are inside a loop (per the compiler's own "in Loop"
block annotations of the gfx950 assembly) versus in straight-line prologue / epilogue code.

    python scripts/scratch_report.py > profiles/r04/scratch_report.txt

No GPU needed (hipcc -S --cuda-device-only). A spill outside every loop is paid once per launch; one
inside a loop is paid per iteration and is what matters for the latency-bound kernels."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nlsolver_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def report(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", tmp.name, src], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().splitlines()
    rows, func, in_loop = {}, None, False
    for ln in lines:
        m = re.match(r"^(_ZN4nlsg\w+):", ln)
        if m:
            func, in_loop = m.group(1), False
            rows[func] = {"ops": 0, "loop_ops": 0}
            continue
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb."):
            in_loop = "Loop" in t
        if func and t.startswith("scratch_"):
            rows[func]["ops"] += 1
            rows[func]["loop_ops"] += in_loop
        m = re.match(r"^; (NumVgprs|ScratchSize|Occupancy): (\d+)", t)
        if m and func:
            rows[func][m.group(1)] = int(m.group(2))
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    print(f"{'kernel':78s} {'VGPRs':>5s} {'scratch B/lane':>14s} {'scratch ops':>11s} {'in loops':>8s}")
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        rows = {k: v for k, v in report(os.path.join(CSRC, f)).items() if v.get("ScratchSize", 0) > 0}
        names = demangle(list(rows))
        for k, v in rows.items():
            name = re.sub(r"\(.*", "", names[k]).replace("void nlsg::", "").replace("nlsg::", "")
            print(f"{name[:78]:78s} {v.get('NumVgprs', 0):5d} {v['ScratchSize']:14d} {v['ops']:11d} {v['loop_ops']:8d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
