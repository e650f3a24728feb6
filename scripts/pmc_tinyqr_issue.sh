#!/bin/bash
# scripts/pmc_tinyqr_issue.sh — on the GPU box: rocprofv3 SQ counters of tinyqr_lm_kernel
# (bench.py --workload tinyqr: 8192 systems of 576 x 64): which unit its 638 steps keep busy.
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_tinyqr
mkdir -p $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  NLSG_BENCH_NO_CONSISTENCY_CHECK=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o tq -- python3 $root/bench.py --workload tinyqr --steps 2 --no-cpu-baseline > /dev/null 2> $out/$tag.err || echo "failed $set"
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
    if "tinyqr" in k: print(k, {c: round(v["mean"]) for c, v in sorted(d.items())})
PY
