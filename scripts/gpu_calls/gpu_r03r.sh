#!/bin/bash
# call 18: attention kernel A/B on ONE box: previous commit's library (build/prev) vs this tree; stamps; the executor / TP tests
o=gpurun_out/r03r; mkdir -p $o; : > $o/decoder.txt
for i in 1 2; do
  echo "== previous kernel" >> $o/decoder.txt; timeout 200 build/prev/bench_decoder 32 2048 128 2 2>&1 | grep "ctx" >> $o/decoder.txt
  echo "== this tree" >> $o/decoder.txt; timeout 200 build/bench_decoder 32 2048 128 2 2>&1 | grep "ctx" >> $o/decoder.txt
done
timeout 200 build/probe/bench_decoder 32 2048 128 1 2>&1 | grep "attention kernel" >> $o/decoder.txt
cut -c1-300 $o/decoder.txt
export EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_tp_gpu.py tests/test_sampler.py -q -k "not perplexity and not end_to_end and not ring_stream and not real_layer" 2>&1 | grep -v amdgpu.ids | tail -30 > $o/tests.txt; tail -5 $o/tests.txt
