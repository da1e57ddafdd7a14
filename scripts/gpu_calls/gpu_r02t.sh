#!/bin/bash
set -u
OUT=gpurun_out/r02t
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cp exllama_amd/libexl_amd.so /tmp/lib_nocap.so; timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
run() {  # tag model-args
  tag=$1; shift
  timeout 700 python bench.py --model "$@" --steps 2 --warmup 1 --no-cpu-baseline --prompt 2048 > $OUT/bench_$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["decode_best_tokens_per_s"], {k:v["us_per_launch"] for k,v in d["roofline"]["classes"].items()}, d["roofline"]["frac"])
except Exception as e: print("$tag", "ERR", e)
PY
}
for variant in nocap; do
  if [ $variant = cap ]; then cp build/cap/libexl_amd.so exllama_amd/libexl_amd.so; else cp /tmp/lib_nocap.so exllama_amd/libexl_amd.so; fi
  run ${variant}_13b 13b
  run ${variant}_13bact 13b --act-order
  run ${variant}_33b 33b --groupsize 32 --act-order
  run ${variant}_65b 65b
  run ${variant}_7b 7b
done
cp /tmp/lib_nocap.so exllama_amd/libexl_amd.so
