#!/bin/bash
# round 5, call 8: the whole GPU suite INCLUDING the full-depth perplexity case (as the driver will run it), smoke(), and the default
# bench line with all its sub-runs
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 1800 python -m pytest tests -q -m gpu --durations=12 > $OUT/full_tests.log 2>&1; echo "suite rc $?" >> $OUT/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -n 22 $OUT/full_tests.log; tail -n 2 $OUT/smoke.log
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("7B", d["value"], d["prefill_tokens_per_s"], d["decode_best_tokens_per_s"], d["logits_finite"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:160])
    for k, v in (d.get("other_configs") or {}).items():
        print(k, v.get("value"), v.get("prefill_tokens_per_s"), v.get("decode_best_tokens_per_s"), v.get("logits_finite"), v.get("seconds"), v.get("error"))
    dr = d.get("dropin_reference_model_py") or {}
    print("dropin", dr.get("decode_worst_tokens_per_s"), dr.get("decode_best_tokens_per_s"), dr.get("prefill_tokens_per_s"), dr.get("seconds"), dr.get("error"))
except Exception as e:
    print("bench ERR", e)
PY
