#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_tp_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r02x_tp.txt
EXL_BENCH_FORCE_DIST=1 EXL_TP_ALWAYS_COLLECTIVE=1 MASTER_PORT=29533 timeout 600 python bench.py --tensor-parallel --no-cpu-baseline --layers 8 > gpurun_out/r02x_tp1_rccl.json 2> gpurun_out/r02x_tp1_rccl.err
cat gpurun_out/r02x_tp.txt; tail -c 400 gpurun_out/r02x_tp1_rccl.json; tail -3 gpurun_out/r02x_tp1_rccl.err
