#!/bin/bash
# call 15: tensor-parallel generation (greedy / sampled on every rank), the multi-GPU bench modes with their per-rank report on one rank over RCCL
o=gpurun_out/r03o; mkdir -p $o
timeout 900 python -m pytest tests/test_tp_gpu.py tests/test_sampler.py -q 2>&1 | grep -v amdgpu.ids | tail -25 > $o/tests.txt; tail -5 $o/tests.txt
EXL_BENCH_FORCE_DIST=1 MASTER_PORT=29511 timeout 600 python bench.py --tensor-parallel --gpus 1 --steps 2 --warmup 1 > $o/bench_tp_1gpu.json 2> $o/tp.err; tail -c 900 $o/bench_tp_1gpu.json; tail -3 $o/tp.err
EXL_BENCH_FORCE_DIST=1 MASTER_PORT=29512 timeout 600 python bench.py --layer-split --gpus 1 --steps 2 --warmup 1 > $o/bench_ls_1gpu.json 2> $o/ls.err; head -c 600 $o/bench_ls_1gpu.json; tail -3 $o/ls.err
