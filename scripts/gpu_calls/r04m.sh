#!/bin/bash
# round 4, call 14+: the prompt-pass attention kernel on its own (scripts/bench_flash.hip): check, time, phase probe
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for b in ${FLASH_BINS:-bench_flash bench_flash_probe}; do
  echo "== $b" >> $OUT/flash.log
  timeout 120 build/$b 2048 32 32 0 20 >> $OUT/flash.log 2>&1; echo "rc $?" >> $OUT/flash.log
done
timeout 120 build/bench_flash 1920 40 40 0 10 >> $OUT/flash.log 2>&1
timeout 120 build/bench_flash 300 8 2 77 5 >> $OUT/flash.log 2>&1
cat $OUT/flash.log
