#!/bin/bash
set -u
OUT=gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout=900 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 600 python bench.py --layer-split --gpus 1 --steps 2 --warmup 1 > $OUT/bench_layer_split.json 2> $OUT/bench_layer_split.err; echo "exit $?" >> $OUT/bench_layer_split.err
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/bench.err
tail -25 $OUT/pytest_gpu.log; tail -3 $OUT/bench_layer_split.err; head -c 1500 $OUT/bench_layer_split.json; echo; head -c 700 $OUT/bench.json
