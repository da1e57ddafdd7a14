#!/bin/bash
# round 6, call 27: prompts of up to 32 rows with the narrow kernel (the decode GEMV's shape: 8 waves split K, activations in registers) for
# EVERY launch (EXL_GEMM_SMALL_NARROW=1) against the wide kernel's one- / two-row-tile shapes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
EXL_GEMM_SMALL_NARROW=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts or batched" 2>&1 | tail -4
for p in 2 8 16 32; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/wide: /" >> $OUT/short_prompt.txt
  EXL_GEMM_SMALL_NARROW=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/narrow: /" >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
EXL_GEMM_SMALL_NARROW=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p16 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 16 > /dev/null 2> $OUT/prof16.err
find $OUT/prof -name "p16_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt16_narrow.csv \;
grep -E "t16g|t16r|to_frag" $OUT/kernel_stats_prompt16_narrow.csv | cut -c1-170
find $OUT -type f ! -name "*stats*" -size +2M -delete
