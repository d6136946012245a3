#!/bin/bash
# round 3, experiment 3: layer-chain kernel on/off and its crossover in tiles per workgroup
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_exp3"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
for mt in 0 512 1024 2048; do
  TAG=chain_max_tiles_$mt ARDAE_CHAIN_MAX_TILES=$mt python scratch/exp_shard.py 64 128 256 512 2>&1 | grep ms/step >> "$O/times.txt"
done
cat "$O/times.txt"
python scratch/stamp_shard.py 64 2>&1 | grep -v amdgpu.ids | tail -16
python -m pytest tests/test_cdae_gpu.py tests/test_engine_gpu.py tests/test_dp_gpu.py -x -q -m gpu > "$O/tests.log" 2>&1 || { tail -40 "$O/tests.log"; exit 1; }
tail -3 "$O/tests.log"
