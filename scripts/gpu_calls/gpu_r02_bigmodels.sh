#!/bin/bash
# the other BASELINE shapes with the round's final binary: one protocol pass each (2 timed steps)
mkdir -p gpurun_out/r02big
run() {
  tag=$1; shift
  timeout 900 python bench.py --model "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02big/bench_$tag.json 2> gpurun_out/r02big/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02big/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["decode_best_tokens_per_s"], d["prefill_tokens_per_s"], d["roofline"]["frac"], d["prefill_roofline"]["frac"])
except Exception as e: print("$tag", "ERR", e)
PY
}
run 13b 13b
run 13bact 13b --act-order
run 65b 65b
