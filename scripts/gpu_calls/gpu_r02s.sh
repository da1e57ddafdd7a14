#!/bin/bash
set -u
OUT=gpurun_out/r02s
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4
for m in "13b" "13b --act-order" "33b --groupsize 32 --act-order" "65b" "7b"; do
  tag=$(echo $m | tr -d ' -')
  timeout 700 python bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$tag.json").read().strip().splitlines()[-1])
    print("$m", d["value"], d["decode_best_tokens_per_s"], d["prefill_tokens_per_s"], d["path_roofline"]["decode_worst"]["frac_of_8TBps"], d["path_roofline"]["decode_best"]["frac_of_8TBps"], {k:v["us_per_launch"] for k,v in d["roofline"]["classes"].items()}, d["roofline"]["frac"])
except Exception as e: print("$m", "ERR", e)
PY
done
