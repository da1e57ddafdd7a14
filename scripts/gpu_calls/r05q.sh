#!/bin/bash
# round 5, call 17: same box, alternating: the ring kernels with the wave-private activation image (product) and with the block-strided copy +
# image barrier of rounds 3-4 (-DEXL_RING_BLOCK_IMAGE build) through scripts/bench_decoder.cpp
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== wave-private image (product), run $rep" >> $OUT/ab.txt
  timeout 100 build/bench_decoder 32 2048 128 2 2>&1 | grep -v logits >> $OUT/ab.txt
  echo "== block image + barrier (rounds 3-4), run $rep" >> $OUT/ab.txt
  timeout 100 build/ring_blockimg/bench_decoder 32 2048 128 2 2>&1 | grep -v logits >> $OUT/ab.txt
done
cut -c1-200 $OUT/ab.txt
