#!/bin/bash
# the dual (gate/up) prefill GEMM on random, constant and zero activations: same instruction stream, different switching activity
mkdir -p gpurun_out
{
for d in random const_x zero_x; do
  echo "== $d"; DUAL_DATA=$d timeout 300 python scripts/bench_dual.py
done
rocm-smi --showpower --showclocks 2>&1 | head -30
} > gpurun_out/r02z4.txt 2>&1
grep -v amdgpu.ids gpurun_out/r02z4.txt
