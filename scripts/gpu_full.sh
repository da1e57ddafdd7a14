#!/bin/bash
# the round-end checks the driver runs: every GPU test, smoke(), the default bench line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/full_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/full_smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
cat gpurun_out/full_tests.txt; tail -2 gpurun_out/full_smoke.txt; tail -c 600 gpurun_out/full_bench.json
