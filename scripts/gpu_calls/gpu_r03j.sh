#!/bin/bash
# call 10: kernel arguments of dec_ring_kernel requested in one batch -- A/B on bench_decoder + phase stamps + bitwise tests
mkdir -p gpurun_out/r03j
o=gpurun_out/r03j/decoder.txt
: > $o
echo "== ring defaults" >> $o; timeout 200 build/bench_decoder 32 2048 128 2 2>&1 | grep -v amdgpu.ids >> $o
echo "== ring defaults, run 2" >> $o; timeout 200 build/bench_decoder 32 2048 128 2 2>&1 | grep -v amdgpu.ids >> $o
echo "== stream (EXL_DEC_RING=0)" >> $o; EXL_DEC_RING=0 timeout 200 build/bench_decoder 32 2048 128 2 2>&1 | grep -v amdgpu.ids >> $o
echo "== phase stamps" >> $o; timeout 200 build/ring_probe/bench_decoder 32 2048 128 1 2>&1 | grep -v amdgpu.ids >> $o
cat $o | cut -c1-250
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "ring_stream or real_layer_shapes or executor_matches" 2>&1 | tail -5 | tee gpurun_out/r03j/tests.txt
