#!/bin/bash
# A/B of two builds of the library on the same box: scratch/ab.sh <libA> <libB>
for rep in 1 2; do
for lib in "$@"; do
  for e in 0 1 2; do echo -n "$(basename $lib) "; ARDAE_LIB=$lib EPI=$e python scratch/bench_linear.py 2>&1 | tail -1; done
done
done
