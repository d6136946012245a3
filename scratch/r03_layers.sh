#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
timeout -k 10 600 python -m pytest tests/test_linear_gpu.py -x -q -m gpu -k "wide_layers" 2>&1 | tail -n 6
timeout -k 10 600 python -m pytest tests/test_cdae_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "nrow or production or full_size or additivity" 2>&1 | tail -n 4
for rep in 1 2 3; do
  for gb in 512 256; do
    for v in "all_on" "ARDAE_WIDE_LAYERS=0"; do
      if [ "$v" = "all_on" ]; then e="X=1"; else e="$v"; fi
      echo -n "B=$gb $v: "; env $e BENCH_GLOBAL_B=$gb python bench.py --steps 150 --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
    done
  done
done
