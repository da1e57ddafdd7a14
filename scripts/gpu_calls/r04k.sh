#!/bin/bash
# round 4, call 12: LoRA operands at one row through the executor's launches (op level): parity and the reference-style token loop with an adapter
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "adapter or lora or fused or binding" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
tail -n 6 $OUT/t.log; tail -n 1 $OUT/lora.log | cut -c1-1400
