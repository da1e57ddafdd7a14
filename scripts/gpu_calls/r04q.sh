#!/bin/bash
# round 4: the short-context attention kernel with DPP reductions: decode parity tests, then the benchmark line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "decode or executor or greedy or sample or smoke or attn" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -n 4 $OUT/t.log; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("prefill_tokens_per_s","decode_worst_tokens_per_s","decode_best_tokens_per_s","prefill_ms")})
PY
