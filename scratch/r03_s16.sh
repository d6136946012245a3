#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
for rep in 1 2; do
  for gb in 256 512; do
    for v in "all_on" "ARDAE_SMALL16_MAX_BLOCKS=128" "ARDAE_SMALL16_MAX_BLOCKS=512"; do
      if [ "$v" = "all_on" ]; then e="X=1"; else e="$v"; fi
      echo -n "B=$gb $v: "; env $e BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
    done
  done
done
