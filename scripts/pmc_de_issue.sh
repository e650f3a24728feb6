#!/bin/bash
# scripts/pmc_de_issue.sh — on the GPU box: rocprofv3 SQ issue counters of the DE headline run
# (vector / scalar instructions per wave, how busy each unit is, how long the waves wait), three
# --pmc passes of `bench.py --steps 20` under gpurun_out/pmc_de/. What DESIGN.md §3 "What bounds
# the generation at pop = 65 536" quotes.
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_de
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o de -- python3 $root/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-north-star --no-other-configs > /dev/null 2> $out/$tag.err || echo "failed $set"
  find $out/$tag -type f ! -name '*counter_collection.csv' -delete
done
ls -R $out | head -30
