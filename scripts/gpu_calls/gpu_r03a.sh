#!/bin/bash
# Round 3, GPU call 1: the rolling-ring decode stream (decode_ring.hip) -- parity (whole GPU suite, incl. the bit-for-bit A/B test
# against dec_stream_kernel and the new cold-launch stress test of the hand-counted GEMM kernels), then per-class times of the 7B
# decode step with the compiler-scheduled stream, the ring, and the ring without its start-up barrier; the default bench line.
mkdir -p gpurun_out
o=gpurun_out/r03a
mkdir -p $o
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu 2>&1 | tail -25 > $o/tests_model.txt
cat $o/tests_model.txt
timeout 1200 python -m pytest tests -q -m gpu --ignore=tests/test_model_gpu.py 2>&1 | tail -25 > $o/tests_rest.txt
cat $o/tests_rest.txt
for cfg in "0 1" "1 1" "1 0"; do
    set -- $cfg
    echo "== EXL_DEC_RING=$1 EXL_DEC_RING_FENCE=$2" | tee -a $o/decoder_ab.txt
    EXL_DEC_RING=$1 EXL_DEC_RING_FENCE=$2 timeout 300 build/bench_decoder 32 2048 128 2>&1 | grep -v amdgpu.ids | tee -a $o/decoder_ab.txt
done
echo "== EXL_DEC_RING=1 EXL_DEC_NSPLIT=8" | tee -a $o/decoder_ab.txt
EXL_DEC_NSPLIT=8 timeout 300 build/bench_decoder 32 2048 128 2>&1 | grep -v amdgpu.ids | tee -a $o/decoder_ab.txt
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err
tail -c 1500 $o/bench.json
