#!/bin/bash
# after the short-prompt GEMM: op + model parity, then the bench line (other_lengths carries the 128-token prompt)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r02w_tests.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02w_bench.json 2> gpurun_out/r02w_bench.err
cat gpurun_out/r02w_tests.txt; tail -c 1500 gpurun_out/r02w_bench.json
