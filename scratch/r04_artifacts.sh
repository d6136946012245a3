#!/bin/bash
# Round-4 profile artefacts (copied / condensed into profiles/ afterwards): bash scratch/r04_artifacts.sh
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
O="$R/gpurun_out/r04_art"; rm -rf "$O"; mkdir -p "$O"
python bench.py --steps 100 --warmup 20 > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench default done"
export ARDAE_DEBUG_KNOBS=1
for gb in 256 128 64; do
  BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$O/bench_b$gb.json" 2> "$O/bench_b$gb.err"
done
echo "shards done"
python scratch/bench_configs.py 1 2 4 5 6 7 8 9 10 > "$R/gpurun_out/r04_configs.txt" 2> "$O/configs.err"
echo "configs done"
unset ARDAE_DEBUG_KNOBS
bash scripts_profile.sh r04 > "$O/profile.log" 2>&1
echo "profile passes done"
for c in 1 4 5 9; do bash scratch/r04_kstats_cfg.sh r04cfg$c $c > "$O/cfg$c.txt" 2>&1; echo "cfg $c done"; done
cd /tmp && export TMPDIR=/tmp
ARDAE_DEBUG_KNOBS=1 BENCH_GLOBAL_B=64 rocprofv3 --kernel-trace --output-format csv -d "$O/tr_b64" -- python3 "$R/bench.py" --steps 60 --warmup 10 --prof-steps 0 --no-cpu-baseline > /dev/null 2> "$O/tr_b64.log"
cd "$R" && python tools/step_timeline.py "$O/tr_b64" > "$O/timeline_b64.txt" 2>&1
echo "timeline done"
