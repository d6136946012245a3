#!/bin/bash
# per-rank shards of the 1 / 2 / 4 / 8-GPU runs on ONE GPU (round 4): bash scratch/r04_shards.sh
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
mkdir -p gpurun_out/r04_shards
for gb in 512 256 128 64; do
  BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r04_shards/bench_b$gb.json 2> gpurun_out/r04_shards/bench_b$gb.err
  python -c "import json; d=json.load(open('gpurun_out/r04_shards/bench_b$gb.json')); print('B=$gb', round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
done
