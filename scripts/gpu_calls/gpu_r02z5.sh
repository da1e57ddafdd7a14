#!/bin/bash
# shader clock and package power WHILE the dual GEMM runs back to back (rocm-smi sampled once a second), random vs zero activations
mkdir -p gpurun_out
{
for d in random zero_x; do
  echo "== dual GEMM, $d activations, 15000 launches back to back"
  DUAL_DATA=$d DUAL_REPS=15000 timeout 300 python scripts/bench_dual.py &
  pid=$!
  sleep 4
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|Power|GPU use|mclk"; sleep 1; done
  wait $pid
done
echo "== idle"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power"
rocm-smi --showmaxpower 2>/dev/null | grep -i power
} > gpurun_out/r02z5.txt 2>&1
grep -v amdgpu.ids gpurun_out/r02z5.txt
