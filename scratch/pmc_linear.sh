#!/bin/bash
# usage: bash scratch/pmc_linear.sh <tag> ; env ARDAE_LIB/EPI/MROWS honoured
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITERS=3
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/scratch/bench_linear.py > /dev/null 2> $OUT/a.log
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_ANY --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/scratch/bench_linear.py > /dev/null 2> $OUT/b.log
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "linear_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("$TAG", k)
            for c, xs in sorted(v.items()):
                print("   %-32s %14.0f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
