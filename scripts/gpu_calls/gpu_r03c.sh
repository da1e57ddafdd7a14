#!/bin/bash
# Round 3, GPU call 3: the flat shallow ring (U <= 4 loads in flight per lane, wait counts replayed at compile time): parity (model
# tests incl. the bit-for-bit A/B against dec_stream_kernel at 7B / 13B / 65B / 70B shapes), then 7B per-class times by ring depth,
# the phase stamps at depth 2 and 4, and the loads-only variant.
mkdir -p gpurun_out
o=gpurun_out/r03c
mkdir -p $o
timeout 900 python -X faulthandler -m pytest tests/test_model_gpu.py -x -q -m gpu > $o/tests_model_full.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|error\|Fatal\|fault" $o/tests_model_full.txt | tail -12
run() { echo "== $1" | tee -a $o/decoder_ab.txt; shift; env "$@" 2>&1 | grep -v amdgpu.ids | tee -a $o/decoder_ab.txt; }
run "stream (EXL_DEC_RING=0)"      EXL_DEC_RING=0 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4"                 EXL_DEC_RING_DEPTH=4 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3"                 EXL_DEC_RING_DEPTH=3 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 2"                 EXL_DEC_RING_DEPTH=2 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 2, no fence"       EXL_DEC_RING_DEPTH=2 EXL_DEC_RING_FENCE=0 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4, phase stamps"   EXL_DEC_RING_DEPTH=4 timeout 300 build/ring_probe/bench_decoder 32 2048 128
run "ring depth 2, phase stamps"   EXL_DEC_RING_DEPTH=2 timeout 300 build/ring_probe/bench_decoder 32 2048 128
run "ring depth 4, loads only"     EXL_DEC_RING_DEPTH=4 timeout 300 build/ring_ablate/bench_decoder 32 2048 128
run "ring depth 2, loads only"     EXL_DEC_RING_DEPTH=2 timeout 300 build/ring_ablate/bench_decoder 32 2048 128
