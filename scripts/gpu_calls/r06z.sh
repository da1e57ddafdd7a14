#!/bin/bash
# round 6, call 35: prompts of up to 32 rows with two row-blocks per step in the wide kernel (PF 3) against one (EXL_GEMM_T16G_RBS=1)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -k "frag or short_prompts or batched" 2>&1 | tail -6
for p in 2 8 16 32; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/two row-blocks per step: /" >> $OUT/short_prompt.txt
  EXL_GEMM_T16G_RBS=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/one (EXL_GEMM_T16G_RBS=1): /" >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p16 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 16 > /dev/null 2> $OUT/prof16.err
find $OUT/prof -name "p16_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt16.csv \;
grep -E "t16g|t16r|to_frag" $OUT/kernel_stats_prompt16.csv | cut -c1-170
find $OUT -type f ! -name "*stats*" -size +2M -delete
