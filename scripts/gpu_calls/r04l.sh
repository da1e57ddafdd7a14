#!/bin/bash
# round 4, call 13: the x A launch of an adapter beside the GEMV launch (side stream / parallel graph branch): A/B and parity
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "adapter or lora" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
EXL_DEC_LORA_FORK=0 timeout 600 python scripts/bench_lora.py --out $OUT/lora_nofork.json > $OUT/lora_nofork.log 2>&1
timeout 600 python scripts/bench_lora.py --out $OUT/lora_fork.json > $OUT/lora_fork.log 2>&1
tail -n 4 $OUT/t.log; for f in nofork fork; do python - <<PY
import json
d=json.load(open("$OUT/lora_$f.json"))
print("$f", {k:v.get("decode_tokens_per_s") for k,v in d.items() if isinstance(v,dict) and "decode_tokens_per_s" in v})
PY
done
