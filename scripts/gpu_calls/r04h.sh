#!/bin/bash
# round 4, call 9: LoRA kernels with every load requested up front: parity + kernel stats + full-model numbers
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lora or adapter" > $OUT/t_model.log 2>&1; echo "model rc $?" >> $OUT/t_model.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/scripts/bench_lora.py --layers 8 > $OUT/lora8.log 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
tail -n 3 $OUT/t_model.log; grep "lora" $OUT/kernel_stats.csv | cut -c1-140; tail -n 1 $OUT/lora.log | cut -c1-1400
