#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
timeout -k 10 300 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "two_ranks_on_one_gpu" 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -n 30
for rep in 1 2; do
  for gb in 64 128 512; do
    for v in "all_on" "ARDAE_SMALL_CHAIN=0"; do
      if [ "$v" = "all_on" ]; then e="X=1"; else e="$v"; fi
      echo -n "B=$gb $v: "; env $e BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
    done
  done
done
