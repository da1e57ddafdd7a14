#!/bin/bash
# round 4, call 2: traffic-aware tail split, compiled binding (parity + drop-in timing A/B), perplexity / TP tests, bench tiers
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export EXL_TOL_STATS=$OUT/tol_stats.jsonl
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or dual or qkv or prompt_fusions or binding or attn or mlp" > $OUT/t_ops.log 2>&1; echo "ops rc $?" >> $OUT/t_ops.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_tp_gpu.py tests/test_reference_dropin_gpu.py -q -m gpu -k "perplexity or tensor_parallel or dropin or reference or smoke or lora" > $OUT/t_model.log 2>&1; echo "model rc $?" >> $OUT/t_model.log
unset EXL_TOL_STATS
timeout 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_7b.json 2> $OUT/bench_7b.err
timeout 500 python bench.py --model 13b --act-order --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b_act_gptq.json 2> $OUT/bench_13b_act_gptq.err
timeout 400 python scripts/bench_dropin.py --out $OUT/dropin_fast.json --profile $OUT/dropin_fast_profile.txt > $OUT/dropin_fast.log 2>&1
EXL_NO_FAST_BINDING=1 timeout 400 python scripts/bench_dropin.py --out $OUT/dropin_ctypes.json > $OUT/dropin_ctypes.log 2>&1
tail -n 3 $OUT/t_ops.log; tail -n 3 $OUT/t_model.log
tail -n 4 $OUT/dropin_fast.log $OUT/dropin_ctypes.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d.get("prefill_tokens_per_s"), d.get("decode_best_tokens_per_s"), (d.get("prefill_roofline") or {}).get("avg_launch_us"), d.get("host_argmax_loop"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
