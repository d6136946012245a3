#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
for v in 0 1 2 4 8 14; do
  lib=""; [ "$v" != "0" ] && lib="$PWD/scratch/lib_x9_$v.so"
  echo -n "X9_EXP=$v: "
  ARDAE_LIB=$lib python bench.py --steps 100 --warmup 20 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms')"
done
