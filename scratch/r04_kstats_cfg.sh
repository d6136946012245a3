#!/bin/bash
# Per-kernel table of one of scratch/bench_configs.py's configurations: bash scratch/r04_kstats_cfg.sh <tag> <config number>
set -eu
TAG="${1:?tag}"; CFG="${2:?config}"
R="${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ARDAE_DEBUG_KNOBS=1 ARDAE_GRAPH=0 CFG_STEPS=5
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT"/trace -- python3 "$R/scratch/bench_configs.py" $CFG > "$OUT"/bench.txt 2> "$OUT"/trace.log
cd "$R" && python3 tools/summarize_profile.py "$OUT" "gpurun_out/$TAG" > /dev/null
cat "$OUT"/bench.txt | grep "^config"; head -16 "gpurun_out/${TAG}_kernel_stats.csv"
