#!/bin/bash
# scripts/pmc_ubench.sh <binary> <tag> [args] — rocprofv3 SQ counter passes of a ubench binary
# (per-kernel means into gpurun_out/pmc_<tag>/summary.json and stdout).
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
bin=$root/$1; tag=$2; shift 2
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$t -o u -- $bin "$@" > $out/$t.out 2> $out/$t.err || echo "failed $set"
  find $out/$t -type f ! -name '*counter_collection.csv' -delete
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open("$out/summary.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print(k, {c: round(v["mean"]) for c, v in sorted(d.items())})
PY
