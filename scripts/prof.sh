#!/bin/bash
# scripts/prof.sh TAG BENCH_ARGS... — per-kernel durations of one bench.py run on the GPU box:
# rocprofv3 --kernel-trace --stats, summary copied to gpurun_out/TAG_kernel_stats.csv and the bench
# line to gpurun_out/TAG_bench.json (the files that get committed under profiles/rNN/).
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- \
  python3 "$root/bench.py" "$@" > "$root/gpurun_out/${tag}_bench.json" 2> "$root/gpurun_out/${tag}_bench.err"
rc=$?
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
exit $rc
