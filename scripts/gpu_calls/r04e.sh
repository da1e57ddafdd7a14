#!/bin/bash
# round 4, call 5: fold fixes (identity store map, LoRA test), skinny LoRA down-projection, tile order A/B again (alternating, more steps)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "half_matmul or lora or binding" > $OUT/t_ops.log 2>&1; echo "ops rc $?" >> $OUT/t_ops.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_tp_gpu.py -q -m gpu -k "folded or lora or real_layer_shapes or tensor_parallel or ring_stream" > $OUT/t_model.log 2>&1; echo "model rc $?" >> $OUT/t_model.log
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
for rep in 1 2; do
  for mode in new old; do
    if [ $mode = old ]; then export EXL_GEMM_TILE_ROWS=0; else unset EXL_GEMM_TILE_ROWS; fi
    timeout 500 python bench.py --model 13b --act-order --steps 4 --warmup 1 --no-cpu-baseline --gen 16 --no-roofline-probe > $OUT/bench_13b_act_${mode}_$rep.json 2> $OUT/bench_13b_act_${mode}_$rep.err
  done
done
unset EXL_GEMM_TILE_ROWS
timeout 500 python bench.py --model 13b --act-order --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b_act_full.json 2> $OUT/bench_13b_act_full.err
tail -n 3 $OUT/t_ops.log; tail -n 4 $OUT/t_model.log; tail -n 2 $OUT/lora.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "prefill", d.get("prefill_tokens_per_s"), d.get("prefill_ms"), "decode", d["value"], d.get("decode_best_tokens_per_s"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
