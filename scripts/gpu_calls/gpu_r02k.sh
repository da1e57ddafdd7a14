#!/bin/bash
set -u
OUT=gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout=900 --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/bench.err
tail -22 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
for k in ("value","decode_best_tokens_per_s","prefill_tokens_per_s","other_lengths","prefill_roofline","cpu_baseline"): print(k, d.get(k))
print({k:v["us_per_launch"] for k,v in d["roofline"]["classes"].items()}, d["roofline"]["frac"], d["roofline"].get("traffic_source"))
PY
