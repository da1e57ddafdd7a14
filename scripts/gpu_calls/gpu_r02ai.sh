#!/bin/bash
# software-pipelined dual GEMM (q4_gemm_t16d2_kernel): parity (bit-identical to the separate products), sustained A/B, step probe
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dual or full_size" 2>&1 | tail -5 > gpurun_out/r02ai_tests.txt
{
echo "== pipelined (default)"; DUAL_REPS=2000 timeout 300 python scripts/bench_dual.py
echo "== predecessor"; EXL_GEMM_DUAL_UNPIPELINED=1 DUAL_REPS=2000 timeout 300 python scripts/bench_dual.py
echo "== pipelined again"; DUAL_REPS=2000 timeout 300 python scripts/bench_dual.py
echo "== step probe (probe build, its own data)"; timeout 200 build/probe_dual 2048
} > gpurun_out/r02ai_ab.txt 2>&1
cat gpurun_out/r02ai_tests.txt; grep -v amdgpu.ids gpurun_out/r02ai_ab.txt
