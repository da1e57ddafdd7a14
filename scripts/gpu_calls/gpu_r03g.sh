#!/bin/bash
# Round 3, GPU call 7: the whole GPU suite with the round's new tests (tolerances 4e-3 with RMS / block criteria -- measured
# ratios logged to tol_stats.jsonl --, real-shape end-to-end prefill vs the oracle, perplexity to 2 dp, tensor parallel vs the
# oracle, make_q4 double-call guard, HIP embedding / head, cold launches incl. split-K), the 257-512 row GEMM routes, the drop-in
# path timed, the CPU baseline at full depth, the other BASELINE shapes.
mkdir -p gpurun_out
o=gpurun_out/r03g
mkdir -p $o
EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl timeout 2400 python -X faulthandler -m pytest tests -q -m gpu > $o/tests_full.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|error\|Fatal\|fault" $o/tests_full.txt | tail -20
timeout 600 bash scripts/gpu_r03_first.sh > $o/tile128.txt 2>&1
tail -12 $o/tile128.txt
timeout 900 python scripts/bench_dropin.py --out $o/dropin.json > $o/dropin.txt 2>&1
tail -3 $o/dropin.txt
timeout 900 python bench.py --cpu-baseline-full > $o/bench_full_cpu.json 2> $o/bench_full_cpu.err
python - <<PY
import json
try:
    d=json.loads(open("$o/bench_full_cpu.json").read().strip().splitlines()[-1]); print("7b", d["value"], d["decode_best_tokens_per_s"], d["prefill_tokens_per_s"], d["cpu_baseline"])
except Exception as e: print("bench ERR", e)
PY
run() {
  tag=$1; shift
  timeout 900 python bench.py --model "$@" --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_$tag.json 2> $o/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$o/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["decode_best_tokens_per_s"], d["prefill_tokens_per_s"], d["roofline"]["frac"], d["prefill_roofline"]["frac"], d["path_roofline"]["decode_worst"]["frac_of_8TBps"])
except Exception as e: print("$tag", "ERR", e)
PY
}
run 13b 13b
run 13bact 13b --act-order
run 33bg32act 33b --groupsize 32 --act-order
run 65b 65b
run 70b 70b
