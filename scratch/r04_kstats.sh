#!/bin/bash
# Quick per-kernel table of the contract workload on the GPU box: bash scratch/r04_kstats.sh <tag> [GLOBAL_B]
# (rocprofv3 --kernel-trace of bench.py with eager launches; condensed by tools/summarize_profile.py into gpurun_out/<tag>_kernel_stats.csv)
set -eu
TAG="${1:-r04k}"; GB="${2:-512}"
R="${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ARDAE_DEBUG_KNOBS=1 ARDAE_GRAPH=0 BENCH_GLOBAL_B=$GB
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT"/trace -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 > "$OUT"/bench_under_rocprof.json 2> "$OUT"/trace.log
cd "$R" && python3 tools/summarize_profile.py "$OUT" "gpurun_out/$TAG" > /dev/null
head -28 "gpurun_out/${TAG}_kernel_stats.csv"
