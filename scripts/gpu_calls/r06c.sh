#!/bin/bash
# round 6, call 3: what the fused launch's activation vector looks like next to the two-launch one (1 layer)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export EXL_DEC_ENGINE_SPINS=20000
ENGINE_VARIANT=1 timeout 90 build/bench_decoder 2 2048 128 1 2 2>&1 | grep -E "engine check|two-launch|per-launch|-- engine" > $OUT/engine.txt
cut -c1-300 $OUT/engine.txt
