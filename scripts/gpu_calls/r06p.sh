#!/bin/bash
# round 6, call 19: why is the drop-in's 1920-token prompt pass 174 ms (56.7 k tokens/s in round 5, 11.1 k now)?  kernel stats of scripts/bench_dropin.py
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/bench_dropin.py --layers 8 --out $OUT/dropin8.json 2>&1 | tail -3
cat $OUT/dropin8.json | cut -c1-600
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o d -- python $GRAFT_REPO_ROOT/scripts/bench_dropin.py --layers 8 > /dev/null 2> $OUT/prof.err
find $OUT/prof -name "d_kernel_stats*" -exec cp {} $OUT/kernel_stats_dropin.csv \;
head -14 $OUT/kernel_stats_dropin.csv | cut -c1-190
find $OUT -type f ! -name "*stats*" -size +2M -delete
