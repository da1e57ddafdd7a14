#!/bin/bash
# round 4: the 8-wave attention kernel inside the library: attention / prompt tests, then the 7B benchmark line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "attention or flash or prefill or perplexity or smoke or real_shape" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -n 5 $OUT/t.log; cut -c1-900 $OUT/bench.json
