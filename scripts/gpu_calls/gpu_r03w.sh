#!/bin/bash
# call 24 (round-3 final measurements): bench.py as the driver runs it, the same command under rocprofv3 --stats, PMC traffic, smoke, the 13B test
o=gpurun_out/r03w; mkdir -p $o
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $o/bench_7b.json 2> $o/bench.err; tail -c 400 $o/bench_7b.json
bash scripts/gpu_r03_profiles.sh > $o/profiles.log 2>&1; tail -16 $o/profiles.log | cut -c1-200
cp -r gpurun_out/r03prof/kernel_stats.csv gpurun_out/r03prof/pmc_traffic.json gpurun_out/r03prof/bench_under_rocprof.json $o/ 2>/dev/null
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "real_layer_shapes and 13b" 2>&1 | tail -2
