#!/bin/bash
# First GPU call of the next round (about 2 GPU-minutes): is the 128 x 128 tile GEMM clean now that the final wait of its K loop is
# tied to the in-flight registers?  The failing scenario was "first launch after an idle period, freshly copied
# activation tensor, uninitialised output", one to two failures per process before the fix.
mkdir -p gpurun_out
out=gpurun_out/r03_tile128_validation.txt
: > $out
for i in $(seq 1 8); do
    echo "== process $i" >> $out
    EXL_GEMM_TILE128=1 timeout 60 python scripts/diag_tile128.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out
done
echo "processes with a bad case: $(grep -c ' bad [1-9]' $out) lines; faults: $(grep -c 'Memory access fault' $out)" | tee -a $out
EXL_GEMM_TILE128=1 timeout 60 build/probe_tile128 /tmp/ref.bin 400 4096 11008 32 | tail -1 | tee -a $out
EXL_GEMM_TILE128=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -2 | tee -a $out
# the split-K form on top (off by default): parity of the GEMM tests, then 7B-layer times at 300 / 384 / 512 rows:
# 256-row kernels (default) | 128-row tile | 128-row tile with K cut in two
EXL_GEMM_TILE128=1 EXL_GEMM_SPLITK=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "gemm" 2>&1 | tail -2 | tee -a $out
for m in 300 384 512; do
    echo "rows $m: default $(timeout 60 build/bench_gemm $m 50 | tail -1)" | tee -a $out
    echo "rows $m: tile128 $(EXL_GEMM_TILE128=1 timeout 60 build/bench_gemm $m 50 | tail -1)" | tee -a $out
    echo "rows $m: tile128 + split-K $(EXL_GEMM_TILE128=1 EXL_GEMM_SPLITK=1 timeout 60 build/bench_gemm $m 50 | tail -1)" | tee -a $out
done
