#!/bin/bash
# Round 3, GPU call 2: which model test crashed in call 1 (full log this time), then the decoder A/B that call 1 lost to a
# bench_decoder built for the wrong architecture: compiler-scheduled stream | ring | ring without its start-up barrier | KV splits
# 8 | three blocks per CU (one unit per block) | the ring's phase stamps | the ring's loads without the arithmetic.
mkdir -p gpurun_out
o=gpurun_out/r03b
mkdir -p $o
timeout 900 python -X faulthandler -m pytest tests/test_model_gpu.py -x -v -m gpu > $o/tests_model_full.txt 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|fault\|Fault\|Abort\|passed\|failed" $o/tests_model_full.txt | tail -40
run() { echo "== $1" | tee -a $o/decoder_ab.txt; shift; env "$@" 2>&1 | grep -v amdgpu.ids | tee -a $o/decoder_ab.txt; }
run "stream (EXL_DEC_RING=0)"      EXL_DEC_RING=0 timeout 300 build/bench_decoder 32 2048 128
run "ring"                         timeout 300 build/bench_decoder 32 2048 128
run "ring, no start-up barrier"    EXL_DEC_RING_FENCE=0 timeout 300 build/bench_decoder 32 2048 128
run "ring, 8 KV splits"            EXL_DEC_NSPLIT=8 timeout 300 build/bench_decoder 32 2048 128
run "ring, 3 blocks per CU"        EXL_DEC_BLOCKS_PER_CU=3 timeout 300 build/bench_decoder 32 2048 128
run "ring, phase stamps"           timeout 300 build/ring_probe/bench_decoder 32 2048 128
run "ring, loads only"             timeout 300 build/ring_ablate/bench_decoder 32 2048 128
run "stream again"                 EXL_DEC_RING=0 timeout 300 build/bench_decoder 32 2048 128
