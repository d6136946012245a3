#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
export ARDAE_DEBUG_KNOBS=1
for v in "ARDAE_SMALL_FAST=0" "ARDAE_SMALL16_MAX_BLOCKS=0" "ARDAE_FUSED_A1=0" "ARDAE_SMALL_CHAIN=0"; do
  echo "== $v"
  env $v timeout -k 10 400 python -m pytest tests/test_training_quality_gpu.py -q -s -m gpu -k "config2_widths" 2>&1 | grep -E "config-#2 widths|passed|failed"
done
