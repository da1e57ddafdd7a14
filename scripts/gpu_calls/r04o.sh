#!/bin/bash
# round 4: the 8-wave attention kernel against the 4-wave one, alternating on one box (scripts/bench_flash.hip)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for shape in "2048 32 32 0 20" "1920 40 40 0 20" "2048 52 52 0 10" "2048 64 8 0 10" "512 32 32 0 20" "128 32 32 0 20" "300 8 2 77 20" "64 32 32 1984 20"; do
    echo "== $shape" >> $OUT/ab.log
    echo -n "8-wave: " >> $OUT/ab.log; timeout 120 build/bench_flash $shape 2>&1 | tr '\n' ' ' >> $OUT/ab.log; echo >> $OUT/ab.log
    echo -n "4-wave: " >> $OUT/ab.log; EXL_FLASH_4WAVE=1 timeout 120 build/bench_flash $shape 2>&1 | tr '\n' ' ' >> $OUT/ab.log; echo >> $OUT/ab.log
  done
done
grep -o "heads.*past [0-9]*\|flash prefill: [0-9.]* us\|^[48]-wave\|NaN [0-9]*\|ref| [0-9.e+-]*" $OUT/ab.log | paste -sd' ' | sed 's/8-wave/\n8-wave/g; s/4-wave/\n  4-wave/g' | head -60
