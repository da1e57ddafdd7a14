#!/bin/bash
# short-prompt GEMM: parity first, then A/B against the 128x128 tile kernel at M = 32 .. 1024
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or threshold or lora or attn_2" 2>&1 | tail -5 > gpurun_out/r02u_tests.txt
{
for M in 32 64 128 256 512 1024; do
  echo "== M $M skinny"; EXL_GEMM_SKINNY_MAX=2048 timeout 120 build/bench_gemm $M 50
  echo "== M $M tile kernel"; EXL_GEMM_SKINNY_MAX=0 timeout 120 build/bench_gemm $M 50
done
for M in 64 128 256; do for C in 0 1 2 3; do echo "== M $M skinny cfg $C"; EXL_GEMM_SKINNY_CFG=$C EXL_GEMM_SKINNY_MAX=2048 timeout 120 build/bench_gemm $M 50; done; done
} > gpurun_out/r02u_gemm.txt 2>&1
cat gpurun_out/r02u_tests.txt; grep -v "^M " gpurun_out/r02u_gemm.txt
