#!/bin/bash
# round 3, experiment 2: the plan-of-linear-graphs engine (two linear graphs on two streams instead of one forked graph)
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_exp2"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
TAG=plan_graphs python scratch/exp_shard.py 64 128 256 512 > "$O/times.txt" 2>&1
TAG=plan_eager ARDAE_GRAPH=0 python scratch/exp_shard.py 64 128 >> "$O/times.txt" 2>&1
TAG=plan_graph_no_overlap ARDAE_OVERLAP=0 python scratch/exp_shard.py 64 128 >> "$O/times.txt" 2>&1
cat "$O/times.txt"
python -m pytest tests/test_engine_gpu.py tests/test_dp_gpu.py tests/test_surface_gpu.py -x -q -m gpu > "$O/tests.log" 2>&1 || { tail -40 "$O/tests.log"; exit 1; }
tail -3 "$O/tests.log"
cd /tmp && export TMPDIR=/tmp
TAG=trace_plan rocprofv3 --kernel-trace --output-format csv -d "$O/tr_plan" -- python3 "$R/scratch/exp_shard.py" 64 > "$O/tr_plan.log" 2>&1
tail -n 2 "$O/tr_plan.log"
