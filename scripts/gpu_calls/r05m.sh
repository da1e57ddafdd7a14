#!/bin/bash
# round 5, call 13: the round-end check as the driver runs it -- whole GPU suite (full-depth perplexity through the golden log-likelihoods),
# smoke(), default bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $OUT/full_tests.log 2>&1; echo "suite rc $?" >> $OUT/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -n 14 $OUT/full_tests.log; tail -n 2 $OUT/smoke.log
python - <<PY
import json
try:
    lines = [l for l in open("$OUT/bench_default.json").read().strip().splitlines() if l.strip()]
    print("stdout lines:", len(lines), "last is json:", lines[-1].startswith("{"))
    d = json.loads(lines[-1])
    print("7B", d["value"], d["prefill_tokens_per_s"], d["decode_best_tokens_per_s"], d["logits_finite"], d["roofline"]["frac"], d["ms_per_step"])
    for k, v in (d.get("other_configs") or {}).items():
        print(k, v.get("value"), v.get("prefill_tokens_per_s"), v.get("decode_best_tokens_per_s"), v.get("logits_finite"), v.get("seconds"), v.get("error"))
    dr = d.get("dropin_reference_model_py") or {}
    print("dropin", dr.get("decode_worst_tokens_per_s"), dr.get("decode_best_tokens_per_s"), dr.get("prefill_tokens_per_s"), dr.get("seconds"), dr.get("error"))
except Exception as e:
    print("bench ERR", e)
PY
