#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_dropin_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r02aa.txt
cat gpurun_out/r02aa.txt
