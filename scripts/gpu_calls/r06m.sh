#!/bin/bash
# round 6, call 16: one / two row tiles per block for prompts of a few rows (chat turns): parity of every shape, timing at 2 .. 128 rows
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -60 > $OUT/tests_frag_ops.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_frag_ops.txt
EXL_TOL_STATS=$OUT/tol_short.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -30 > $OUT/tests_short.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_short.txt
for p in 2 8 16 17 32 33 64 128; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" >> $OUT/short_prompt.txt
done
for p in 8 16 32; do
  EXL_GEMM_NO_FRAG=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/op by op (EXL_GEMM_NO_FRAG=1): /' >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
for p in 16 32; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p$p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt $p > /dev/null 2> $OUT/prof$p.err
  find $OUT/prof -name "p${p}_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt$p.csv \;
  echo "== prompt $p"; grep -E "t16g|t16r|to_frag|rope_qk|flash|attention|attn" $OUT/kernel_stats_prompt$p.csv | cut -c1-170
done
find $OUT -type f ! -name "*stats*" -size +2M -delete
