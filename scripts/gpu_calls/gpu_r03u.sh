#!/bin/bash
# call 21: chunked attention kernel for the multi-split buckets + round-2 kernel for the one-split bucket: same-box A/B vs the previous library, tests
o=gpurun_out/r03u; mkdir -p $o; : > $o/decoder.txt
for v in prev/bench_decoder bench_decoder prev/bench_decoder bench_decoder; do echo "== $v" >> $o/decoder.txt; timeout 100 build/$v 32 2048 128 2 2>&1 | grep "graph replay\|per-launch" >> $o/decoder.txt; done
cut -c1-200 $o/decoder.txt
export EXL_TOL_STATS=$PWD/$o/tol_stats.jsonl
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_tp_gpu.py tests/test_sampler.py tests/test_reference_dropin_gpu.py -q -k "not perplexity and not end_to_end and not ring_stream" 2>&1 | grep -v amdgpu.ids | tail -30 > $o/tests.txt; tail -5 $o/tests.txt
