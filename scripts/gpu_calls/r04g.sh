#!/bin/bash
# round 4, call 8: where the time of a token with an adapter inside the graph goes (rocprofv3 kernel stats of scripts/bench_lora.py, 8 layers)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/scripts/bench_lora.py --layers 8 > $OUT/lora8.log 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
tail -n 1 $OUT/lora8.log | cut -c1-900
head -30 $OUT/kernel_stats.csv | cut -c1-150
