#!/bin/bash
# scripts/profile_round.sh ROUND — on the GPU box: for every benchmark workload one
# `rocprofv3 --kernel-trace --stats` run (per-kernel durations + the bench line) and two PMC
# passes, FETCH_SIZE and WRITE_SIZE apart (they do not fit one pass, MI355X_MICROARCH.md), under
# gpurun_out/prof_ROUND/. Summarised afterwards, off the box, by profiles/summarize_pmc.py into
# profiles/ROUND/{pmc_summary,de_pmc_summary}.json; the csv / json files are copied there as they are.
# scripts/profile_round.sh ROUND "tag tag ..." limits the run to those workloads;
# NLSG_PROFILE_NO_PMC=1 skips the two counter passes (a refresh of the kernel-stats csv alone).
round=${1:-r04}
only=" ${2:-} "
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$round
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  local tag=$1; shift
  if [ "$only" != "  " ] && [[ "$only" != *" $tag "* ]]; then return 0; fi
  echo "== $tag: $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o "$tag" -- \
    python3 "$root/bench.py" "$@" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err" || return 1
  cp "$(find "$out/${tag}_stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
  [ -n "${NLSG_PROFILE_NO_PMC:-}" ] && return 0
  for ctr in FETCH_SIZE WRITE_SIZE; do
    NLSG_BENCH_NO_CONSISTENCY_CHECK=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/${tag}_$ctr" -o "$tag" -- \
      python3 "$root/bench.py" "$@" --steps 20 --warmup 2 --no-cpu-baseline --no-north-star --no-other-configs \
      > /dev/null 2> "$out/${tag}_$ctr.err" || return 1
    # keep only the counter csv (the traces are large)
    find "$out/${tag}_$ctr" -type f ! -name '*counter_collection.csv' -delete
  done
}
run de_c2 --steps 1000 --warmup 50 --no-other-configs --no-north-star &&
run de_ns --pop-per-gpu 1048576 --steps 100 --warmup 5 --no-cpu-baseline &&
run pso_accel --workload pso-accel --steps 100 --warmup 10 &&
run pso_vanilla --workload pso-vanilla --steps 100 --warmup 10 --no-cpu-baseline &&
run bfgs --workload bfgs &&
run bfgs_sym --workload bfgs --bfgs-symmetric --no-cpu-baseline &&
run bfgs_ref --workload bfgs --bfgs-reference-order --no-cpu-baseline &&
run lm --workload lm &&
run lm_qr --workload lm --lm-solver qr --no-cpu-baseline &&
run nm --workload nm && run sann --workload sann && run nmpso --workload nmpso &&
run bfgs_fd --workload bfgs-fd && run lm_fd --workload lm-fd &&
run lm_n128 --workload lm --lm-n 128 && run lm_n256 --workload lm --lm-n 256 &&
run lm_n512 --workload lm --lm-n 512 --no-cpu-baseline && run lm_n1024 --workload lm --lm-n 1024 --no-cpu-baseline &&
run tinyqr --workload tinyqr
echo "rc=$?"
du -sh "$out"
