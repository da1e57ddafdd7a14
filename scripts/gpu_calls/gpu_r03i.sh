#!/bin/bash
# call 9: the tests reworked after call 7 (conditioning-aware bounds, real-shape perplexity, e2e prefill) + split-K default
mkdir -p gpurun_out/r03i
export EXL_TOL_STATS=$PWD/gpurun_out/r03i/tol_stats.jsonl
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_cold_launch_gpu.py tests/test_ops_gpu.py -q -k "executor or perplexity or end_to_end or cold or gemm" 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r03i/tests.txt
tail -5 gpurun_out/r03i/tests.txt
