#!/bin/bash
# round 6, call 21: LoRA inside the executor with the adapters' x @ A launches on a side stream beside the GEMV launch (a parallel branch of
# the captured graph): parity (the LoRA tests), rate at rank 16 / 64 against the serial order (EXL_DEC_LORA_SERIAL=1)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_sampler.py -q -k "lora or sampler" 2>&1 | tail -12 > $OUT/tests_lora.txt
cat $OUT/tests_lora.txt
timeout 600 python scripts/bench_lora.py --out $OUT/lora_beside.json > /dev/null 2> $OUT/lora1.err
EXL_DEC_LORA_SERIAL=1 timeout 600 python scripts/bench_lora.py --out $OUT/lora_serial.json > /dev/null 2> $OUT/lora2.err
python - <<PY
import json
for f in ("lora_beside", "lora_serial"):
    d = json.load(open("$OUT/%s.json" % f))
    print(f, {k: (v.get("decode_tokens_per_s"), v.get("decode_vs_no_adapter_same_path"), v.get("prefill_tokens_per_s")) for k, v in d.items() if isinstance(v, dict)})
PY
