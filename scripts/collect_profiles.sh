#!/bin/bash
# scripts/collect_profiles.sh ROUND — off the GPU box: copy what scripts/profile_round.sh ROUND left
# under gpurun_out/prof_ROUND into profiles/ROUND (kernel-stats csv and bench line per workload) and
# summarise the PMC passes (profiles/summarize_pmc.py) into pmc_summary.json / de_pmc_summary.json.
round=${1:-r04}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gpurun_out/prof_$round
dst=$root/profiles/$round
mkdir -p "$dst"
for f in "$src"/*_kernel_stats.csv "$src"/*_bench.json; do [ -f "$f" ] && cp "$f" "$dst/"; done
rm -f "$dst/pmc_summary.json" "$dst/de_pmc_summary.json"
args=()
for d in "$src"/*_FETCH_SIZE "$src"/*_WRITE_SIZE; do
  [ -d "$d" ] || continue
  tag=$(basename "$d" | sed 's/_FETCH_SIZE$//; s/_WRITE_SIZE$//')
  case $tag in de_c2|de_ns) ;; *) args+=("$tag=$d");; esac
done
python3 "$root/profiles/summarize_pmc.py" "$dst/pmc_summary.json" "${args[@]}"
python3 "$root/profiles/summarize_pmc.py" "$dst/de_pmc_summary.json" \
  c2b="$src/de_c2_FETCH_SIZE" c2b="$src/de_c2_WRITE_SIZE" nsb="$src/de_ns_FETCH_SIZE" nsb="$src/de_ns_WRITE_SIZE"
[ -f "$root/gpurun_out/pmc_pso/summary.json" ] && cp "$root/gpurun_out/pmc_pso/summary.json" "$dst/pso_issue_summary.json"
ls "$dst" | wc -l
