#!/bin/bash
# round 6, call 13: would the wide kernel's <4, 2> shape (128 blocks) beat the narrow kernel on o_proj / down_proj at 128 rows?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
for mb in 160 100; do
  for p in 128 64 250; do
    EXL_GEMM_T16G_MIN_BLOCKS=$mb timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/min blocks $mb: /" >> $OUT/short_prompt.txt
  done
done
cat $OUT/short_prompt.txt
cd /tmp
EXL_GEMM_T16G_MIN_BLOCKS=100 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p128 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 128 > /dev/null 2> $OUT/prof128.err
find $OUT/prof -name "p128_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt128_mb100.csv \;
grep -E "t16g|t16r|to_frag" $OUT/kernel_stats_prompt128_mb100.csv | cut -c1-170
find $OUT -type f ! -name "*stats*" -size +2M -delete
