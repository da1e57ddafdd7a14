#!/bin/bash
# Round 3, GPU call 4: DPP reductions in the activation prologues, 16-wave blocks for the one-block-per-CU launches, attention
# chunk scaled with the wave count.  Parity (model tests), then 7B per-class times over: ring depth x blocks per CU x 16-wave
# blocks, 8 KV splits with 8-wave attention blocks, phase stamps.
mkdir -p gpurun_out
o=gpurun_out/r03d
mkdir -p $o
timeout 900 python -X faulthandler -m pytest tests/test_model_gpu.py -x -q -m gpu > $o/tests_model_full.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|error\|Fatal\|fault" $o/tests_model_full.txt | tail -12
run() { echo "== $1" | tee -a $o/decoder_ab.txt; shift; env "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^logits" | tee -a $o/decoder_ab.txt; }
run "stream (EXL_DEC_RING=0)"          EXL_DEC_RING=0 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4, wide"               EXL_DEC_RING_DEPTH=4 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4, no wide"            EXL_DEC_RING_DEPTH=4 EXL_DEC_RING_WIDE=0 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3, wide"               EXL_DEC_RING_DEPTH=3 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3, wide, 3 blocks/CU"  EXL_DEC_RING_DEPTH=3 EXL_DEC_BLOCKS_PER_CU=3 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4, wide, 3 blocks/CU"  EXL_DEC_RING_DEPTH=4 EXL_DEC_BLOCKS_PER_CU=3 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 2, wide, 3 blocks/CU"  EXL_DEC_RING_DEPTH=2 EXL_DEC_BLOCKS_PER_CU=3 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3, wide, 8 KV splits x 8 waves" EXL_DEC_RING_DEPTH=3 EXL_DEC_NSPLIT=8 EXL_DEC_ATTN_WAVES=8 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3, wide, 16 KV splits x 8 waves" EXL_DEC_RING_DEPTH=3 EXL_DEC_ATTN_WAVES=8 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3, wide, phase stamps" EXL_DEC_RING_DEPTH=3 timeout 300 build/ring_probe/bench_decoder 32 2048 128
run "ring depth 3, wide, 3 blocks/CU, phase stamps" EXL_DEC_RING_DEPTH=3 EXL_DEC_BLOCKS_PER_CU=3 timeout 300 build/ring_probe/bench_decoder 32 2048 128
