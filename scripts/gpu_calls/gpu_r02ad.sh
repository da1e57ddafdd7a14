#!/bin/bash
# 7B-shape decoder, groupsize 128 vs 32, ablation levels of the stream kernels (EXL_DEC_ABLATE_BUILD library):
# 0 = complete, 1 = no dequant + MFMA, 3 = also no scale / zero loads, 4 = also no activation loads
mkdir -p gpurun_out
{
for gs in 128 32; do for a in 0 1 3; do
  echo "== groupsize $gs ablate $a"; EXL_DEC_ABLATE=$a timeout 300 build/abl/bench_decoder 8 2048 $gs 2>&1 | grep "per-launch" | head -1
done; done
} > gpurun_out/r02ad.txt 2>&1
cat gpurun_out/r02ad.txt
