#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests/test_linear_gpu.py tests/test_cdae_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -n 5
for rep in 1 2 3; do
  for gb in 64 128; do
    echo -n "B=$gb: "; BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
  done
done
python scratch/bench_configs.py 6 7 8 2>&1 | grep "^config"
