#!/bin/bash
# call 17: attention kernel with chunk-interleaved splits (position-independent first requests), DPP reductions: parity, A/B, stamps
o=gpurun_out/r03q; mkdir -p $o
export EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_tp_gpu.py -q -k "not perplexity and not end_to_end and not ring_stream" 2>&1 | grep -v amdgpu.ids | tail -30 > $o/tests.txt; tail -5 $o/tests.txt
for i in 1 2; do timeout 200 build/bench_decoder 32 2048 128 2 2>&1 | grep -v amdgpu.ids | grep "ctx" >> $o/decoder.txt; done
timeout 200 build/probe/bench_decoder 32 2048 128 1 2>&1 | grep -v amdgpu.ids | grep "attention kernel" >> $o/decoder.txt
cut -c1-300 $o/decoder.txt
