#!/bin/bash
# A/B on ONE device: the pieces of the round-3 small-shard work switched off one at a time (devices of the pool differ by several %)
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_ab_shard"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
for rep in 1 2; do
TAG=all_on python scratch/exp_shard.py 64 128 256 512 2>&1 | grep ms/step >> "$O/times.txt"
TAG=no_chain_kernel ARDAE_CHAIN_MAX_TILES=0 python scratch/exp_shard.py 64 128 2>&1 | grep ms/step >> "$O/times.txt"
TAG=no_fused_draw ARDAE_FUSED_DRAW=0 python scratch/exp_shard.py 64 128 512 2>&1 | grep ms/step >> "$O/times.txt"
TAG=no_overlap ARDAE_OVERLAP=0 python scratch/exp_shard.py 64 128 512 2>&1 | grep ms/step >> "$O/times.txt"
TAG=aql_batching_off DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python scratch/exp_shard.py 64 2>&1 | grep ms/step >> "$O/times.txt"
TAG=eager ARDAE_GRAPH=0 python scratch/exp_shard.py 64 2>&1 | grep ms/step >> "$O/times.txt"
done
cat "$O/times.txt"
