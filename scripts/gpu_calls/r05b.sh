#!/bin/bash
# round 5, call 2: the fused LoRA launch (parity + rate), head_dim 100, TP over 4 processes, 65B prompt shapes, the HIP half of the
# full-depth perplexity records (symmetric zeros)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT EXL_SKIP_SLOW=1
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -k "lora or adapter or head_dim_100 or half_matmul or q4_attn or q4_mlp or rope or update_cache or attention" > $OUT/t_lora_hd.log 2>&1; echo "rc $?" >> $OUT/t_lora_hd.log
timeout 600 python -m pytest tests/test_multiproc_gpu.py -q -m gpu -k "tensor_parallel" > $OUT/t_tp.log 2>&1; echo "rc $?" >> $OUT/t_tp.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu -k "65b or 22016" > $OUT/t_65b.log 2>&1; echo "rc $?" >> $OUT/t_65b.log
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
EXL_DEC_LORA_SPLIT=1 timeout 600 python scripts/bench_lora.py --out $OUT/lora_split_pair.json > $OUT/lora_split.log 2>&1
timeout 400 python scripts/ppl_full_depth.py --model 7b --seeds 17,18 --hip-only $OUT/ppl > $OUT/ppl_7b.log 2>&1
timeout 500 python scripts/ppl_full_depth.py --model 13b --act-order --seeds 17 --hip-only $OUT/ppl > $OUT/ppl_13b.log 2>&1
tail -n 5 $OUT/t_lora_hd.log; tail -n 4 $OUT/t_tp.log; tail -n 4 $OUT/t_65b.log
tail -n 1 $OUT/lora.log | cut -c1-900; tail -n 1 $OUT/lora_split.log | cut -c1-600
tail -n 2 $OUT/ppl_7b.log $OUT/ppl_13b.log | cut -c1-500
