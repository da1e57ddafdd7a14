#!/bin/bash
set -u
OUT=gpurun_out/r02b
mkdir -p $OUT gpurun_out/ref_golden
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m oracle.make_ref_golden gpurun_out/ref_golden/ref_ops.npz > $OUT/ref_golden.log 2>&1; echo "ref_golden exit $?" >> $OUT/ref_golden.log
for ew in 0 1; do
  EXL_DEC_EARLY_WEIGHTS=$ew timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_ew$ew.json 2> $OUT/bench_ew$ew.err; echo "bench exit $?" >> $OUT/bench_ew$ew.err
done
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout=900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/ref_golden.log; tail -5 $OUT/pytest_gpu.log
for ew in 0 1; do python - <<PY
import json
d=json.loads(open("$OUT/bench_ew$ew.json").read().strip().splitlines()[-1])
print("early_weights=$ew", d["value"], d["decode_best_tokens_per_s"], d["prefill_tokens_per_s"], {k:v["us_per_launch"] for k,v in d["roofline"]["classes"].items()})
PY
done
