#!/bin/bash
# round 5, call 14: rocprofv3 kernel stats of BASELINE configs[4] (65B) and configs[2] (13B g128 act-order) through the same bench.py protocol
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05n
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof65 -o b -- python $GRAFT_REPO_ROOT/bench.py --brief --model 65b --steps 2 --warmup 1 > $OUT/bench_65b_under_rocprof.json 2> $OUT/prof65.err
find $OUT/prof65 -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats_65b.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof13 -o b -- python $GRAFT_REPO_ROOT/bench.py --brief --model 13b --act-order --steps 2 --warmup 1 > $OUT/bench_13b_actorder_under_rocprof.json 2> $OUT/prof13.err
find $OUT/prof13 -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats_13b_actorder.csv \;
find $OUT -type f -name "*.csv" -size +2M -delete
find $OUT/prof65 $OUT/prof13 -type f ! -name "*stats*" -delete
head -8 $OUT/kernel_stats_65b.csv | cut -c1-150; head -8 $OUT/kernel_stats_13b_actorder.csv | cut -c1-150
