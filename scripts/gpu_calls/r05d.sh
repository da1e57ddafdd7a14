#!/bin/bash
# round 5, call 4: centered nibbles -- activation range of the full-depth models (13B act-order, 65B), head-scale calibration for the
# perplexity text (7B, 13B act-order), and the default bench line with finite-logits flags on every config
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT EXL_SKIP_SLOW=1
timeout 300 python scripts/debug/finite_by_layer.py --model 7b --rows 64,600 --head-scales 1,1.5,2,2.5,3,3.5,4,4.6,6 > $OUT/finite_7b.log 2>&1
timeout 300 python scripts/debug/finite_by_layer.py --model 13b --act-order --rows 4,600 --head-scales 1,1.5,2,2.5,3,3.5,4,4.6,6 > $OUT/finite_13b_act.log 2>&1
timeout 400 python scripts/debug/finite_by_layer.py --model 65b --rows 4,600 > $OUT/finite_65b.log 2>&1
timeout 1500 python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
for f in $OUT/finite_*.log; do echo "== $f"; tail -n 22 $f | cut -c1-400; done
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("7B", d["value"], d["prefill_tokens_per_s"], d["decode_best_tokens_per_s"], d["logits_finite"], d["config"].get("INVALID"))
    for k, v in (d.get("other_configs") or {}).items():
        print(k, v.get("value"), v.get("prefill_tokens_per_s"), v.get("decode_best_tokens_per_s"), v.get("logits_finite"), v.get("seconds"), v.get("error"))
    dr = d.get("dropin_reference_model_py") or {}
    print("dropin", dr.get("decode_worst_tokens_per_s"), dr.get("decode_best_tokens_per_s"), dr.get("prefill_tokens_per_s"), dr.get("seconds"), dr.get("error"))
except Exception as e:
    print("bench ERR", e)
PY
tail -n 5 $OUT/bench_default.err | cut -c1-300
