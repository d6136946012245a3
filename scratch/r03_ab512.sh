#!/bin/bash
# clean A/B of the full-size step (B = 512) on one device: one fresh process per variant, alternating
set -eu
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
for rep in 1 2 3; do
  for v in "all_on" "ARDAE_FUSED_DRAW=0" "ARDAE_OVERLAP=0" "ARDAE_GRAPH=0"; do
    if [ "$v" = "all_on" ]; then e=""; else e="$v"; fi
    echo -n "$v: "; env $e python bench.py --steps 100 --warmup 20 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'steps/s')"
  done
done
