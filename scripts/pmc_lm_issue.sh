#!/bin/bash
# scripts/pmc_lm_issue.sh — on the GPU box: rocprofv3 SQ counters of the LM evaluation kernel
# (bench.py --workload lm): vector vs matrix-core instructions and busy cycles, waits.
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_lm
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $out/mfma_counters.txt
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  NLSG_BENCH_NO_CONSISTENCY_CHECK=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o lm -- python3 $root/bench.py --workload lm --no-cpu-baseline > /dev/null 2> $out/$tag.err || echo "failed $set"
  find $out/$tag -type f ! -name '*counter_collection.csv' -delete
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"dispatches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open("$out/summary.json", "w"), indent=1)
for k, d in res.items():
    if "lm_iter" in k or "qr_step" in k: print(k, {c: (round(v["mean"]), round(v["max"])) for c, v in sorted(d.items())})
PY
cat $out/mfma_counters.txt | tr '\n' ' '
