#!/bin/bash
# Round 3, GPU call 5: ring requests ahead of the activation image (EXL_DEC_RING_PRE), 8-wave attention blocks for the deepest split
# bucket (automatic), parity first.
mkdir -p gpurun_out
o=gpurun_out/r03e
mkdir -p $o
timeout 1200 python -X faulthandler -m pytest tests/test_model_gpu.py -x -q -m gpu > $o/tests_model_full.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|error\|Fatal\|fault" $o/tests_model_full.txt | tail -12
run() { echo "== $1" | tee -a $o/decoder_ab.txt; shift; env "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^logits" | tee -a $o/decoder_ab.txt; }
run "stream (EXL_DEC_RING=0), 16 x 4-wave attention" EXL_DEC_RING=0 EXL_DEC_NSPLIT=16 EXL_DEC_ATTN_WAVES=4 timeout 300 build/bench_decoder 32 2048 128
for d in 4 3; do for p in 0 1 2; do
run "ring depth $d, pre $p"  EXL_DEC_RING_DEPTH=$d EXL_DEC_RING_PRE=$p timeout 300 build/bench_decoder 32 2048 128
done; done
run "ring depth 4, pre 1, no wide" EXL_DEC_RING_DEPTH=4 EXL_DEC_RING_PRE=1 EXL_DEC_RING_WIDE=0 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4, pre 1, phase stamps" EXL_DEC_RING_DEPTH=4 EXL_DEC_RING_PRE=1 timeout 300 build/ring_probe/bench_decoder 32 2048 128
