#!/bin/bash
# Copy / condense what scratch/r04_artifacts.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call): bash scratch/r04_collect.sh
set -eu
cd "$(dirname "$0")/.."
A=gpurun_out/r04_art
last() { grep '^{' "$1" | tail -1; }
last $A/bench_default.json > profiles/r04_bench_default_run.json
for gb in 256 128 64; do last $A/bench_b$gb.json > profiles/r04_bench_${gb}image_shard_run.json; done
python tools/summarize_profile.py gpurun_out/r04 profiles/r04
for c in 1 4 5 9; do cp gpurun_out/r04cfg${c}_kernel_stats.csv profiles/r04_cfg${c}_kernel_stats.csv; done
cp $A/timeline_b64.txt profiles/r04_64image_shard_timeline.txt
# profiles/r04_configs_and_recipes.txt: assembled by hand from two jobs (gpurun_out/r04_configs_a.txt, gpurun_out/r04_configs.txt)
ls -la profiles | grep r04
