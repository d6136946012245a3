#!/bin/bash
# usage: bash scratch/pmc_clock.sh ; env EPI honoured.  Kernel duration (trace) and GRBM_GUI_ACTIVE (cycles) -> clock.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_clock_$EPI
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITERS=20
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/scratch/bench_linear.py > $OUT/run.txt 2> $OUT/a.log
python3 - <<PY
import csv, glob, collections
dur = []
for f in glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_wide" in r["Kernel_Name"] or "linear_kernel" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_wide" in r["Kernel_Name"] or "linear_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = sorted(dur)[len(dur) // 2] if dur else 0
print("EPI $EPI: median kernel duration %.1f us over %d launches" % (d / 1e3, len(dur)))
for c, xs in sorted(acc.items()):
    m = sum(xs) / len(xs)
    print("   %-28s %14.0f" % (c, m), ("-> %.2f GHz (per-XCD cycles / duration)" % (m / 8 / d) if c == "GRBM_GUI_ACTIVE" and d else ""))
PY
cat $OUT/run.txt | tail -1
