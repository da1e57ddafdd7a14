#!/bin/bash
# masked-row MFMA for group sizes 32 / 64 in the decode executor: parity, then the 33B g32 act-order line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_tp_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r02ae_tests.txt
timeout 900 python bench.py --model 33b --groupsize 32 --act-order --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02ae_33b.json 2> gpurun_out/r02ae_33b.err
cat gpurun_out/r02ae_tests.txt
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r02ae_33b.json").read().strip().splitlines()[-1])
    print("33b g32", d["value"], d["decode_best_tokens_per_s"], {k:v["us_per_launch"] for k,v in d["roofline"]["classes"].items()}, d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
