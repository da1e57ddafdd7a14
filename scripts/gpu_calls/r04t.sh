#!/bin/bash
# round 4, after the last kernel changes: rocprofv3 kernel stats of the bench command (the summary profiles/r04_bench_kernel_stats_round_end.csv)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04t
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "prof exit $?" >> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -type f -size +4M -delete
head -n 16 $OUT/kernel_stats.csv | cut -c1-150
