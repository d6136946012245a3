#!/bin/bash
# round 3, experiment 1: where does the 64-image step spend its time (graph replay / eager / capture order), with traces
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_exp1"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
TAG=graph_default python scratch/exp_shard.py 64 128 512 > "$O/times.txt" 2>&1
TAG=graph_side_first ARDAE_SIDE_FIRST=1 python scratch/exp_shard.py 64 128 512 >> "$O/times.txt" 2>&1
TAG=eager ARDAE_GRAPH=0 python scratch/exp_shard.py 64 128 >> "$O/times.txt" 2>&1
TAG=eager_side_first ARDAE_GRAPH=0 ARDAE_SIDE_FIRST=1 python scratch/exp_shard.py 64 128 >> "$O/times.txt" 2>&1
TAG=graph_no_overlap ARDAE_OVERLAP=0 python scratch/exp_shard.py 64 128 >> "$O/times.txt" 2>&1
cat "$O/times.txt"
cd /tmp && export TMPDIR=/tmp
TAG=trace_default rocprofv3 --kernel-trace --output-format csv -d "$O/tr_default" -- python3 "$R/scratch/exp_shard.py" 64 > "$O/tr_default.log" 2>&1
ARDAE_SIDE_FIRST=1 TAG=trace_side_first rocprofv3 --kernel-trace --output-format csv -d "$O/tr_side_first" -- python3 "$R/scratch/exp_shard.py" 64 > "$O/tr_side_first.log" 2>&1
ARDAE_GRAPH=0 TAG=trace_eager rocprofv3 --kernel-trace --output-format csv -d "$O/tr_eager" -- python3 "$R/scratch/exp_shard.py" 64 > "$O/tr_eager.log" 2>&1
tail -2 "$O"/tr_*.log
