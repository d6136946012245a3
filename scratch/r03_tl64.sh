#!/bin/bash
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_tl64"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH_GLOBAL_B=${GB:-64} rocprofv3 --kernel-trace --output-format csv -d "$O/tr" -- python3 "$R/bench.py" --steps 60 --warmup 10 --prof-steps 0 --no-cpu-baseline > "$O/tr.json" 2> "$O/tr.err"
cd "$R"
python tools/step_timeline.py "$O/tr" --out "$O/timeline.txt" > /dev/null
tail -n 30 "$O/timeline.txt"
