#!/bin/bash
# A/B of the fused perturb + sigma + first-DAE-layer kernel on ONE device: fresh process per variant, alternating; 64- and 512-image shards
set -eu
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
export ARDAE_DEBUG_KNOBS=1
O="$R/gpurun_out/r03_ab_a1.txt"; : > "$O"
timeout -k 10 300 python -m pytest tests/test_cdae_gpu.py -x -q -m gpu -k "fused_perturb" 2>&1 | grep -v "^$" | tail -n 25
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "in_kernel_draws or production" 2>&1 | tail -n 5
for rep in 1 2 3; do
  for gb in 64 512; do
    for v in "all_on" "ARDAE_FUSED_A1=0"; do
      if [ "$v" = "all_on" ]; then e="X=1"; else e="$v"; fi
      echo -n "B=$gb $v: " | tee -a "$O"; env $e BENCH_GLOBAL_B=$gb python bench.py --steps 200 --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')" | tee -a "$O"
    done
  done
done
