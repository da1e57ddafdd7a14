#!/bin/bash
# round 5, call 1: the multi-process tests of the real layer-split / tensor-parallel code, decode_path_report tiers, the API-side advisor
# fixes, the HIP half of the full-depth perplexity records, and the default bench line with its sub-runs (other configs, drop-in)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT EXL_SKIP_SLOW=1
export EXL_TOL_STATS=$OUT/tol_stats.jsonl
timeout 900 python -m pytest tests/test_multiproc_gpu.py -q -m gpu > $OUT/t_multiproc.log 2>&1; echo "rc $?" >> $OUT/t_multiproc.log
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "path_report or head_dim_100 or batched_decode or layer_split_runner or perplexity_agrees" > $OUT/t_model.log 2>&1; echo "rc $?" >> $OUT/t_model.log
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_library.py -x -q -m gpu -k "gemm or prompt_fusions or dual or qkv or binding or library" > $OUT/t_ops.log 2>&1; echo "rc $?" >> $OUT/t_ops.log
unset EXL_TOL_STATS
timeout 400 python scripts/ppl_full_depth.py --model 7b --seeds 17,18 --hip-only $OUT/ppl > $OUT/ppl_7b.log 2>&1
timeout 500 python scripts/ppl_full_depth.py --model 13b --act-order --seeds 17 --hip-only $OUT/ppl > $OUT/ppl_13b.log 2>&1
timeout 1500 python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -n 6 $OUT/t_multiproc.log; tail -n 4 $OUT/t_model.log; tail -n 3 $OUT/t_ops.log; tail -n 2 $OUT/ppl_7b.log $OUT/ppl_13b.log | cut -c1-600
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("7B", d["value"], d["prefill_tokens_per_s"], d["decode_best_tokens_per_s"], d["config"]["decode_path_report"]["tier"])
    for k, v in (d.get("other_configs") or {}).items():
        print(k, v.get("value"), v.get("prefill_tokens_per_s"), v.get("decode_best_tokens_per_s"), v.get("seconds"), v.get("error"))
    dr = d.get("dropin_reference_model_py") or {}
    print("dropin", dr.get("decode_worst_tokens_per_s"), dr.get("decode_best_tokens_per_s"), dr.get("prefill_tokens_per_s"), dr.get("seconds"), dr.get("error"))
except Exception as e:
    print("bench ERR", e)
PY
tail -n 5 $OUT/bench_default.err | cut -c1-300
