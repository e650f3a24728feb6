#!/bin/bash
# scripts/pmc_pso_issue.sh — on the GPU box: rocprofv3 SQ issue counters of the Accelerated-PSO move
# kernel (bench.py --workload pso-accel): vector instructions per wave, how busy the vector unit
# is, how long the waves wait on LDS / on anything. Separate --pmc passes under
# gpurun_out/pmc_pso/; summarised (per kernel means) into gpurun_out/pmc_pso/summary.json.
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_pso
mkdir -p $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o pso -- python3 $root/bench.py --workload ${1:-pso-accel} --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/$tag.err || echo "failed $set"
  find $out/$tag -type f ! -name '*counter_collection.csv' -delete
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open("$out/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v["mean"]) for c, v in d.items()})
PY
