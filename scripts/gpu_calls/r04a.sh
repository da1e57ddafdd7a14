#!/bin/bash
# round 4, call 1: parity of the new pieces (truth criterion, shared-map act-order fusions, K-split tails) + first bench lines
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export EXL_TOL_STATS=$OUT/tol_stats.jsonl
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or dual or qkv or prompt_fusions or remap or rms_norm or make_q4" > $OUT/t_ops.log 2>&1; echo "ops rc $?" >> $OUT/t_ops.log
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -k "real_shape or real_layer or executor_matches or golden or token_by_token" > $OUT/t_model.log 2>&1; echo "model rc $?" >> $OUT/t_model.log
timeout 600 python -m pytest tests/test_tp_gpu.py tests/test_cold_launch_gpu.py -q -m gpu -k "tensor_parallel or t16w0 or t16d2" > $OUT/t_tp.log 2>&1; echo "tp rc $?" >> $OUT/t_tp.log
unset EXL_TOL_STATS
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_7b.json 2> $OUT/bench_7b.err
timeout 500 python bench.py --model 13b --act-order --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b_act_gptq.json 2> $OUT/bench_13b_act_gptq.err
timeout 500 python bench.py --model 13b --act-order --act-order-maps independent --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b_act_indep.json 2> $OUT/bench_13b_act_indep.err
timeout 500 python bench.py --model 13b --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b.json 2> $OUT/bench_13b.err
tail -3 $OUT/t_ops.log $OUT/t_model.log $OUT/t_tp.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d.get("prefill_tokens_per_s"), d.get("decode_best_tokens_per_s"), (d.get("prefill_roofline") or {}).get("avg_launch_us"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
