#!/bin/bash
# Usage (on the GPU box, via gpurun): bash scripts_profile.sh <tag>
# Writes rocprofv3 kernel-trace stats and PMC traffic counters for `bench.py` under gpurun_out/<tag>/ ;
# tools/summarize_profile.py condenses them into profiles/.
set -eu
TAG="${1:-r03}"
R="${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"      # rocprofv3 names its files by pid: leftovers of an earlier run under the same tag would be summarised instead
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ARDAE_DEBUG_KNOBS=1 ARDAE_GRAPH=0   # individual launches: the kernel trace and the counters are per kernel either way
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT"/trace -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT"/bench_under_rocprof.json 2> "$OUT"/trace.log
# separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT"/pmc_fetch -- python3 "$R/bench.py" --steps 3 --warmup 2 --prof-steps 1 --no-cpu-baseline > /dev/null 2> "$OUT"/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT"/pmc_write -- python3 "$R/bench.py" --steps 3 --warmup 2 --prof-steps 1 --no-cpu-baseline > /dev/null 2> "$OUT"/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT"/pmc_sq -- python3 "$R/bench.py" --steps 3 --warmup 2 --prof-steps 1 --no-cpu-baseline > /dev/null 2> "$OUT"/pmc_sq.log
ls -R "$OUT" | head -40
