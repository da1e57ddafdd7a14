#!/bin/bash
# Round 3, GPU call 6: the whole GPU suite on the ring defaults (depth 3, 16-wave blocks, 8 x 8-wave attention for the deepest
# bucket, activation requests ahead of the matrix views), stream / ring A/B, the bench line, rocprofv3 stats + PMC traffic.
mkdir -p gpurun_out
o=gpurun_out/r03f
mkdir -p $o
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu > $o/tests_full.txt 2>&1
grep -n "passed\|failed\|FAILED\|Error\|error\|Fatal\|fault" $o/tests_full.txt | tail -12
run() { echo "== $1" | tee -a $o/decoder_ab.txt; shift; env "$@" 2>&1 | grep -v amdgpu.ids | grep -v "^logits" | tee -a $o/decoder_ab.txt; }
run "stream (EXL_DEC_RING=0), 16 x 4-wave attention" EXL_DEC_RING=0 EXL_DEC_NSPLIT=16 EXL_DEC_ATTN_WAVES=4 timeout 300 build/bench_decoder 32 2048 128
run "ring defaults (depth 3)"  timeout 300 build/bench_decoder 32 2048 128
run "ring depth 4"             EXL_DEC_RING_DEPTH=4 timeout 300 build/bench_decoder 32 2048 128
run "ring depth 3, phase stamps" timeout 300 build/ring_probe/bench_decoder 32 2048 128
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err
tail -c 2500 $o/bench.json
bash scripts/gpu_r03_profiles.sh > $o/profiles.txt 2>&1
tail -30 $o/profiles.txt
