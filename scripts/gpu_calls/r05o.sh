#!/bin/bash
# round 5, call 15: phase stamps of the four ring-kernel classes on the current library (scripts/probe_ring.sh build, -DEXL_RING_PROBE)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 build/ring_probe/bench_decoder 32 2048 128 2 > $OUT/ring_probe.txt 2>&1; echo "exit $?" >> $OUT/ring_probe.txt
cut -c1-400 $OUT/ring_probe.txt
