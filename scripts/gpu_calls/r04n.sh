#!/bin/bash
# round 4: ablation builds of the 8-wave attention kernel (timing only; results are wrong by construction)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for b in ${FLASH_BINS}; do
  echo "== $b" >> $OUT/flash.log
  timeout 120 build/$b 2048 32 32 0 20 2>&1 | grep -v "max |out" >> $OUT/flash.log
done
cat $OUT/flash.log
