#!/bin/bash
# round 6, call 7: (a) what the ring kernels would gain from free arithmetic (ablate build: loads without dequantisation + MFMA), same box,
# alternating with the product; (b) the 128-token prompt: time on the stream vs host time to enqueue, and its kernels (rocprofv3)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== product, run $rep" >> $OUT/ablate.txt
  timeout 100 build/bench_decoder 32 2048 128 2 2>&1 | grep -v logits >> $OUT/ablate.txt
  echo "== ring loads without arithmetic (-DEXL_RING_ABLATE), run $rep" >> $OUT/ablate.txt
  timeout 100 build/ring_ablate/bench_decoder 32 2048 128 2 2>&1 | grep -v logits >> $OUT/ablate.txt
done
cut -c1-220 $OUT/ablate.txt
for p in 128 64 256 512; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok" >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p128 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 128 > /dev/null 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt128.csv \;
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
head -25 $OUT/kernel_stats_prompt128.csv | cut -c1-200
