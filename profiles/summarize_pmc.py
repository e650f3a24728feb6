#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass, as MI355X_MICROARCH.md prescribes)
into {tag: {counter: {kernel: {dispatches, mean_KiB}}}}.

    python profiles/summarize_pmc.py OUT.json TAG=DIR [TAG=DIR ...]

DIR is a rocprofv3 output directory holding *_counter_collection.csv; several DIRs may share a
TAG (e.g. the FETCH_SIZE and WRITE_SIZE passes of one workload). FETCH_SIZE / WRITE_SIZE are in
KiB per dispatch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out_path, specs = sys.argv[1], sys.argv[2:]
    result = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for spec in specs:
        tag, d = spec.split("=", 1)
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[row["Counter_Name"]][row["Kernel_Name"]].append(float(row["Counter_Value"]))
        for ctr, kernels in acc.items():
            result.setdefault(tag, {})[ctr] = {
                k: {"dispatches": len(v), "mean_KiB": sum(v) / len(v)}
                for k, v in kernels.items() if k.startswith(("nlsg::", "void nlsg::"))}
    json.dump(result, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
