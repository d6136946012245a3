#!/bin/bash
set -eu
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
for b in 64 128 256 512; do
for d in 0 4; do ARDAE_SC_DEBUG=$d timeout -k 10 120 python scratch/exp_small_chain.py $b 2>&1 | grep "per score"; done
ARDAE_SMALL_CHAIN=0 timeout -k 10 120 python scratch/exp_small_chain.py $b 2>&1 | grep "per score"
done
