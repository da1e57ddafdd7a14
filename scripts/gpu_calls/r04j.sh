#!/bin/bash
# round 4, call 11: the tall-skinny LoRA prompt GEMM: parity and what it does to the prompt pass with an adapter
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "half_matmul or lora or adapter" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
tail -n 4 $OUT/t.log; tail -n 1 $OUT/lora.log | cut -c1-1400
