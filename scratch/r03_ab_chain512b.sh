#!/bin/bash
set -eu
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
for rep in 1 2 3; do
  for v in "ARDAE_CHAIN_MAX_TILES=0 ARDAE_CHAIN_PF_MAX_TILES=0" "ARDAE_CHAIN_MAX_TILES=0 ARDAE_CHAIN_PF_MAX_TILES=4096" "ARDAE_CHAIN_MAX_TILES=4096"; do
    echo -n "B=512 $v: "; env $v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'steps/s')"
  done
done
