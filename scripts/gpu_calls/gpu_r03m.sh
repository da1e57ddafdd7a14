#!/bin/bash
# call 13: the reference's fused ops (q4_attn / q4_attn_2 / q4_mlp) routed through the executor's kernels at one row: parity, then the
# drop-in timing (reference model.py on the shim) fused vs op-by-op on the same box
o=gpurun_out/r03m; mkdir -p $o
export EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_reference_dropin_gpu.py tests/test_model_gpu.py tests/test_sampler.py -q -k "not real_layer_shapes and not ring_stream and not perplexity and not end_to_end" 2>&1 | grep -v amdgpu.ids | tail -40 > $o/tests.txt
tail -6 $o/tests.txt
timeout 600 python scripts/bench_dropin.py --out $o/dropin.json > $o/dropin.txt 2>&1; tail -2 $o/dropin.txt
EXL_OPS_UNFUSED=1 timeout 600 python scripts/bench_dropin.py --out $o/dropin_unfused.json > $o/dropin_unfused.txt 2>&1; tail -2 $o/dropin_unfused.txt
python - <<PY
import json
for t in ("dropin", "dropin_unfused"):
    try:
        d=json.load(open("$o/%s.json" % t)); print(t, d["prefill_tokens_per_s"], d["decode_worst_tokens_per_s"], d["decode_best_tokens_per_s"])
    except Exception as e: print(t, "ERR", e)
PY
