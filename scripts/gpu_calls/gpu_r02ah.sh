#!/bin/bash
# SQ counters of the prefill GEMM kernels (7B shapes, M = 2048): where the MFMA waves' cycles go
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02ah
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DUAL_REPS=4 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_dual.py > $OUT/p1.log 2>&1
DUAL_REPS=4 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p2 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_dual.py > $OUT/p2.log 2>&1
DUAL_REPS=4 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $OUT/p3 -o p -- python $GRAFT_REPO_ROOT/scripts/bench_dual.py > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "t16d" not in r["Kernel_Name"]: continue
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "dispatches", max(len(x) for x in v.values()))
PY
find $OUT -name "*.csv" -size +1M -delete
