#!/bin/bash
# A/B of debug knobs on ONE device: bash scratch/r04_ab.sh "VAR=val ..." "VAR=val ..." ...   (each argument = one variant's environment; "-" = default)
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
GB="${GB:-512}"
for rep in 1 2; do
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e="ARDAE_NOP=1"
  echo -n "B=$GB [$v]: "
  env $e BENCH_GLOBAL_B=$GB python bench.py --steps ${STEPS:-150} --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
done
done
