#!/bin/bash
# round 6, call 1: first run of the fused gate/up -> down launch (decode_engine.hip): bit-identity against the two-launch form, then
# A/B timing in one process (alternating), then the phase stamps; and the tensor-parallel tests with the fp32 all-reduce.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export EXL_DEC_ENGINE_SPINS=20000
echo "== 4 layers, engine check" > $OUT/engine.txt
timeout 90 build/bench_decoder 4 2048 128 1 2 >> $OUT/engine.txt 2>&1
echo "exit $?" >> $OUT/engine.txt
if grep -q "BIT-IDENTICAL" $OUT/engine.txt; then
  echo "== 32 layers, A/B" >> $OUT/engine.txt
  timeout 300 build/bench_decoder 32 2048 128 2 2 >> $OUT/engine.txt 2>&1
  echo "exit $?" >> $OUT/engine.txt
  echo "== 32 layers, phase stamps" >> $OUT/engine.txt
  timeout 120 build/engine_probe/bench_decoder 32 2048 128 1 1 >> $OUT/engine.txt 2>&1
  echo "exit $?" >> $OUT/engine.txt
fi
cut -c1-260 $OUT/engine.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_multiproc_gpu.py tests/test_tp_gpu.py -x -q 2>&1 | tail -15 > $OUT/tp_tests.txt
cat $OUT/tp_tests.txt
