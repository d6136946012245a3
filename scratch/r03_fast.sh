#!/bin/bash
set -eu
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
for b in 64 128 256; do
ARDAE_SC_DEBUG=0 timeout -k 10 120 python scratch/exp_small_chain.py $b 2>&1 | grep "per score"
ARDAE_SMALL_CHAIN=0 timeout -k 10 120 python scratch/exp_small_chain.py $b 2>&1 | grep "per score"
ARDAE_SMALL16_MAX_BLOCKS=0 timeout -k 10 120 python scratch/exp_small_chain.py $b 2>&1 | grep "per score"
done
timeout -k 10 1000 python -m pytest tests/test_linear_gpu.py tests/test_cdae_gpu.py tests/test_engine_gpu.py tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -n 8
O="$GRAFT_REPO_ROOT/gpurun_out/r03_ab_fast.txt"; : > "$O"
for rep in 1 2 3; do
  for gb in 64 128 256; do
    for v in "all_on" "ARDAE_SMALL16_MAX_BLOCKS=0" "ARDAE_SMALL_FAST=0"; do
      if [ "$v" = "all_on" ]; then e="X=1"; else e="$v"; fi
      echo -n "B=$gb $v: " | tee -a "$O"; env $e BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')" | tee -a "$O"
    done
  done
done
