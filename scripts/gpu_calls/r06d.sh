#!/bin/bash
# round 6, call 5: all-zero logits behind correct residual streams in fused mode: which launch?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export EXL_DEC_ENGINE_SPINS=20000
ENGINE_DEBUG=1 ENGINE_VARIANT=1 timeout 90 build/bench_decoder 2 2048 128 1 2 2>&1 | grep -E "engine check|two-launch|debug" > $OUT/engine2.txt
cut -c1-300 $OUT/engine2.txt
