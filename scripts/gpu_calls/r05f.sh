#!/bin/bash
# round 5, call 6: does the tile-structured weight stream run faster through LDS-DMA than through registers? (scripts/bench_ldsdma_stream.hip)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 240 build/bench_ldsdma_stream > $OUT/ldsdma_stream.txt 2>&1; echo "exit $?" >> $OUT/ldsdma_stream.txt
cat $OUT/ldsdma_stream.txt
