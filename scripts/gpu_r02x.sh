#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_tp_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r02x_tp.txt
cat gpurun_out/r02x_tp.txt
