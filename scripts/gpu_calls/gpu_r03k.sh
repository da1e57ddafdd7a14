#!/bin/bash
# call 11: act-order gather inside the ring kernel (PNORM 2): parity first, then the 13B act-order line
o=gpurun_out/r03k; mkdir -p $o
export EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl
timeout 1200 python -m pytest tests/test_model_gpu.py -q -k "ring_stream or real_layer_shapes or executor_matches or golden" 2>&1 | grep -v amdgpu.ids | tail -30 > $o/tests.txt
tail -4 $o/tests.txt
timeout 600 python bench.py --model 13b --act-order --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_13bact.json 2> $o/13bact.err
EXL_DEC_RING=0 timeout 600 python bench.py --model 13b --act-order --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_13bact_stream.json 2> $o/13bact_stream.err
python - <<PY
import json
for t in ("13bact", "13bact_stream"):
    try:
        d=json.loads(open("$o/bench_%s.json" % t).read().strip().splitlines()[-1])
        print(t, d["value"], d["decode_best_tokens_per_s"], d["roofline"]["frac"], d["path_roofline"]["decode_worst"]["frac_of_8TBps"], {k: round(v.get("us", 0), 2) if isinstance(v, dict) else v for k, v in d["roofline"].get("classes", {}).items()})
    except Exception as e: print(t, "ERR", e)
PY
