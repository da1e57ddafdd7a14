#!/bin/bash
set -u
OUT=gpurun_out/r02c
mkdir -p $OUT gpurun_out/ref_golden
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m oracle.make_ref_golden gpurun_out/ref_golden/ref_ops.npz > $OUT/ref_golden.log 2>&1; echo "ref_golden exit $?" >> $OUT/ref_golden.log
timeout 200 build/bench_stream_shape > $OUT/stream_shape.txt 2>&1; echo "exit $?" >> $OUT/stream_shape.txt
for ab in 0 1 2; do
  EXL_DEC_ABLATE=$ab timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_ab$ab.json 2> $OUT/bench_ab$ab.err; echo "bench exit $?" >> $OUT/bench_ab$ab.err
done
tail -3 $OUT/ref_golden.log; cat $OUT/stream_shape.txt
for ab in 0 1 2; do python - <<PY
import json
d=json.loads(open("$OUT/bench_ab$ab.json").read().strip().splitlines()[-1])
print("ablate=$ab", d["value"], d["decode_best_tokens_per_s"], d["prefill_tokens_per_s"], {k:v["us_per_launch"] for k,v in d["roofline"]["classes"].items()})
PY
done
