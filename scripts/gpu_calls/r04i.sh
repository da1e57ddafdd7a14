#!/bin/bash
# round 4, call 10: the round-end checks (every GPU test, smoke(), the default bench line) + the other BASELINE shapes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 1800 python -m pytest tests -q -m gpu > $OUT/full_tests.log 2>&1; echo "suite rc $?" >> $OUT/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench_7b_default.json 2> $OUT/bench_7b_default.err
timeout 500 python bench.py --model 13b --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b.json 2> $OUT/bench_13b.err
timeout 500 python bench.py --model 13b --act-order --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_13b_act.json 2> $OUT/bench_13b_act.err
timeout 700 python bench.py --model 33b --groupsize 32 --act-order --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_33b_g32_act.json 2> $OUT/bench_33b_g32_act.err
timeout 900 python bench.py --model 65b --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_65b.json 2> $OUT/bench_65b.err
tail -n 6 $OUT/full_tests.log; tail -n 2 $OUT/smoke.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "decode", d["value"], d.get("decode_best_tokens_per_s"), "prefill", d.get("prefill_tokens_per_s"), (d.get("path_roofline") or {}).get("prefill", {}).get("frac_of_2.5PF"), (d.get("roofline") or {}).get("frac"), d.get("host_argmax_loop"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
