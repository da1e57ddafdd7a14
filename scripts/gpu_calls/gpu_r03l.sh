#!/bin/bash
# call 12: group sizes 32 / 64 inside the ring kernel (GM): parity first, then the 33B g32 act-order line; 7B g32 through the tiny presets
o=gpurun_out/r03l; mkdir -p $o
export EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl
timeout 1500 python -m pytest tests/test_model_gpu.py -q -k "ring_stream or real_layer_shapes or executor_matches or golden" 2>&1 | grep -v amdgpu.ids | tail -40 > $o/tests.txt
tail -6 $o/tests.txt
timeout 900 python bench.py --model 33b --groupsize 32 --act-order --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_33b.json 2> $o/33b.err
EXL_DEC_RING=0 timeout 900 python bench.py --model 33b --groupsize 32 --act-order --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_33b_stream.json 2> $o/33b_stream.err
python - <<PY
import json
for t in ("33b", "33b_stream"):
    try:
        d=json.loads(open("$o/bench_%s.json" % t).read().strip().splitlines()[-1])
        print(t, d["value"], d["decode_best_tokens_per_s"], d["roofline"]["frac"], d["path_roofline"]["decode_worst"]["frac_of_8TBps"])
    except Exception as e: print(t, "ERR", e)
PY
