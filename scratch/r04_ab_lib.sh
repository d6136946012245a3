#!/bin/bash
# A/B of experiment builds of the library on ONE device: bash scratch/r04_ab_lib.sh <lib.so|-> ...   ("-" = the shipped library)
set -u
cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2; do
for v in "$@"; do
  echo -n "[$v]: "
  if [ "$v" = "-" ]; then unset ARDAE_LIB; else export ARDAE_LIB="$PWD/$v"; fi
  python bench.py --steps ${STEPS:-150} --warmup 30 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'steps/s')"
done
done
