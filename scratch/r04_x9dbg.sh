#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
for v in 0 1 2 4 7; do
  echo -n "ARDAE_X9_DBG=$v: "
  ARDAE_X9_DBG=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms')"
done
