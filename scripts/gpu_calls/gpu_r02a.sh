#!/bin/bash
# round-2 GPU call A: reference-kernel goldens, boundary microbench, GPU tests, short bench
set -u
OUT=gpurun_out/r02a
mkdir -p $OUT gpurun_out/ref_golden
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; free -g | head -2 >> $OUT/gpu.txt
timeout 300 python -m oracle.make_ref_golden gpurun_out/ref_golden/ref_ops.npz > $OUT/ref_golden.log 2>&1; echo "ref_golden exit $?" >> $OUT/ref_golden.log
timeout 120 build/bench_boundary > $OUT/boundary.txt 2>&1; echo "boundary exit $?" >> $OUT/boundary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout=900 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/ref_golden.log; cat $OUT/boundary.txt; tail -2 $OUT/smoke.log; tail -30 $OUT/pytest_gpu.log; head -c 2500 $OUT/bench.json
