#!/bin/bash
# ablation of the short-prompt GEMM at M = 128 (cfg 0 = 64 x 32 blocks, cfg 1 = 32 x 64)
mkdir -p gpurun_out
{
for C in 0 1; do for A in 0 1 2 4 8 3 11 15; do echo "== cfg $C abl $A"; EXL_GEMM_SKINNY_ABL=$A EXL_GEMM_SKINNY_CFG=$C timeout 120 build/bench_gemm 128 50 | head -3; done; done
} > gpurun_out/r02v_abl.txt 2>&1
cat gpurun_out/r02v_abl.txt
