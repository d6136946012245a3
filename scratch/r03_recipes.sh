#!/bin/bash
# rocprofv3 kernel summaries of the five shipped recipes (run_vae_dbmnist.sh proposed-method lines) and of BASELINE configs #1, #4, #5
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_recipes"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python scratch/bench_configs.py 1 2 4 5 6 7 8 9 10 > "$O/configs.txt" 2>&1 || true
cat "$O/configs.txt" | grep config
cd /tmp && export TMPDIR=/tmp
export ARDAE_DEBUG_KNOBS=1 ARDAE_GRAPH=0 CFG_STEPS=5
for c in 1 4 5 6 7 8 9 10; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg$c/trace" -- python3 "$R/scratch/bench_configs.py" $c > "$O/cfg$c.txt" 2> "$O/cfg$c.log" || echo "cfg $c failed"
done
ls "$O"
