#!/bin/bash
# round 5, call 3: where the 13B synthetic model leaves the fp16 range (and whether act-order matters), head-scale calibration of the
# full-depth perplexity text, TP over 4 processes with the measured bound, LoRA pair after the fused kernel's removal
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT EXL_SKIP_SLOW=1
timeout 300 python scripts/debug/finite_by_layer.py --model 13b --act-order --rows 4,64,600 > $OUT/finite_13b_act.log 2>&1
timeout 300 python scripts/debug/finite_by_layer.py --model 13b --rows 4,64,600 > $OUT/finite_13b.log 2>&1
timeout 300 python scripts/debug/finite_by_layer.py --model 13b --act-order --seed 0 --rows 4,600 > $OUT/finite_13b_act_seed0.log 2>&1
timeout 300 python scripts/debug/finite_by_layer.py --model 7b --rows 64,600 --head-scales 4.6,6,7,8,9,10,12 > $OUT/finite_7b.log 2>&1
timeout 600 python -m pytest tests/test_multiproc_gpu.py -q -m gpu -k "tensor_parallel and 4-21" > $OUT/t_tp.log 2>&1; echo "rc $?" >> $OUT/t_tp.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "lora or adapter" > $OUT/t_lora.log 2>&1; echo "rc $?" >> $OUT/t_lora.log
for f in $OUT/finite_*.log; do echo "== $f"; tail -n 12 $f | cut -c1-700; done
tail -n 3 $OUT/t_tp.log $OUT/t_lora.log
