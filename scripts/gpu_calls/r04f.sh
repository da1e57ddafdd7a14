#!/bin/bash
# round 4, call 6: LoRA inside the decode executor (parity + numbers)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lora or adapter" > $OUT/t_model.log 2>&1; echo "model rc $?" >> $OUT/t_model.log
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
tail -n 30 $OUT/t_model.log | cut -c1-250; tail -n 2 $OUT/lora.log
