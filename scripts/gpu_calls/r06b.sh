#!/bin/bash
# round 6, call 2: the fused launch differed from the two-launch form in call 1 (all elements) and ran 22.3 us against 10.4 + 7.0.
# Where: the activation vector (phase A) or the down_proj half; does an agent-scope acquire in front of the activation loads change it
# (variant 3); what the launch costs without the wait (variant 5, results invalid); phase stamps of variants 1 and 5.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export EXL_DEC_ENGINE_SPINS=20000
for v in 1 3; do
  echo "== 1 layer, variant $v" >> $OUT/engine.txt
  ENGINE_VARIANT=$v timeout 90 build/bench_decoder 1 2048 128 1 2 2>&1 | grep -E "engine check|-- engine|per-launch" >> $OUT/engine.txt
  echo "== 4 layers, variant $v" >> $OUT/engine.txt
  ENGINE_VARIANT=$v timeout 90 build/bench_decoder 4 2048 128 1 2 2>&1 | grep -E "engine check|-- engine|per-launch" >> $OUT/engine.txt
done
for v in 1 5; do
  echo "== 32 layers, phase stamps, variant $v" >> $OUT/engine.txt
  timeout 120 build/engine_probe/bench_decoder 32 2048 128 1 $v 2>&1 | grep -E "per-launch|graph replay|fused" >> $OUT/engine.txt
done
cut -c1-420 $OUT/engine.txt
