#!/bin/bash
# round 4: the short-context attention kernel with its position-independent loads issued before the position is read: A/B of two libraries on one box
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== early loads (product), pass $rep" >> $OUT/ab.log
  timeout 200 build/bench_decoder 32 2048 128 2 2>&1 | grep -i "attn\|replay\|tokens/s\|context" >> $OUT/ab.log
  echo "== no early loads, pass $rep" >> $OUT/ab.log
  timeout 200 build/noearly/bench_decoder 32 2048 128 2 2>&1 | grep -i "attn\|replay\|tokens/s\|context" >> $OUT/ab.log
done
cat $OUT/ab.log | cut -c1-200
