#!/bin/bash
# round-3 artefacts for profiles/: default bench line, the 8-rank-shard bench line (64 images on one GPU), its kernel timeline,
# the per-rank shards of the 1/2/4/8-GPU runs, and scripts_profile.sh's rocprofv3 passes of the default bench
set -eu
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_art"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python bench.py --steps 100 --warmup 20 > "$O/bench_default.json" 2> "$O/bench_default.err"
for gb in 256 128 64; do
  BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$O/bench_b$gb.json" 2> "$O/bench_b$gb.err"
done
cd /tmp && export TMPDIR=/tmp
BENCH_GLOBAL_B=64 rocprofv3 --kernel-trace --output-format csv -d "$O/tr_b64" -- python3 "$R/bench.py" --steps 60 --warmup 10 --prof-steps 0 --no-cpu-baseline > "$O/tr_b64.json" 2> "$O/tr_b64.err"
cd "$R"
bash scripts_profile.sh r03 > "$O/profile.log" 2>&1
for f in "$O"/bench_*.json; do python -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], round(d['value'],1), 'steps/s', round(d['ms_per_step'],3), 'ms', 'frac_exec', d.get('whole_step_frac_executed'), 'dominant', r.get('kernel'), r.get('frac'), 'traffic', r.get('traffic'))
" "$f"; done
