#!/bin/bash
# Is the 128 x 128 tile GEMM clean on cold launches, and what do the 257 .. 512-row routes cost?  The validation
# proper is tests/test_cold_launch_gpu.py (28 fresh processes); this script adds the round-2 diagnosis script and 7B-layer times
# at 300 / 384 / 512 rows: 128-row tile (default) | 256-row kernels (EXL_GEMM_NO_TILE128=1) | 128-row tile with K cut in two.
mkdir -p gpurun_out
out=gpurun_out/r03_tile128_validation.txt
: > $out
for i in $(seq 1 4); do
    echo "== process $i" >> $out
    timeout 60 python scripts/diag_tile128.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out
done
echo "processes with a bad case: $(grep -c ' bad [1-9]' $out) lines; faults: $(grep -c 'Memory access fault' $out)" | tee -a $out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -2 | tee -a $out
for m in 300 384 512; do
    echo "rows $m: tile128, whole K $(EXL_GEMM_NO_SPLITK=1 timeout 60 build/bench_gemm $m 50 | tail -1)" | tee -a $out
    echo "rows $m: 256-row kernels $(EXL_GEMM_NO_TILE128=1 timeout 60 build/bench_gemm $m 50 | tail -1)" | tee -a $out
    echo "rows $m: tile128 + split-K (default) $(timeout 60 build/bench_gemm $m 50 | tail -1)" | tee -a $out
done
