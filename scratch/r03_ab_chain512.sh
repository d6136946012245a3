#!/bin/bash
# does the layer-chain kernel (with next-tile staging) pay at 2, 4, 8 tiles per workgroup?  fresh process per variant, one device
set -eu
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
timeout -k 10 300 python -m pytest tests/test_linear_gpu.py -x -q -m gpu -k "chain_kernel" 2>&1 | tail -3
for rep in 1 2; do
  for gb in 512 256 128; do
    for mt in 0 4096; do
      echo -n "B=$gb chain_max_tiles=$mt: "; BENCH_GLOBAL_B=$gb ARDAE_CHAIN_MAX_TILES=$mt python bench.py --steps 100 --warmup 20 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'steps/s')"
    done
  done
done
