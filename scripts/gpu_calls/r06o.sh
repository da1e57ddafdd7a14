#!/bin/bash
# round 6, call 18: the two-steps-deep pipeline of q4_gemm_t16g (EXL_GEMM_T16G_PF=2; slabs two steps ahead in a four-slot ring, weights four
# steps ahead): parity of every block shape on it, timing against PF=1 at 2 .. 256 rows; to_frag with its partial sums requested eight at a time
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
EXL_GEMM_T16G_PF=2 timeout 1500 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -60 > $OUT/tests_frag_ops_pf2.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_frag_ops_pf2.txt
EXL_GEMM_T16G_PF=2 timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -30 > $OUT/tests_short_pf2.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_short_pf2.txt
for p in 2 16 32 64 128 256; do
  for pf in 1 2; do
    EXL_GEMM_T16G_PF=$pf timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/PF=$pf: /" >> $OUT/short_prompt.txt
  done
done
cat $OUT/short_prompt.txt
cd /tmp
for p in 16 128; do
  EXL_GEMM_T16G_PF=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p$p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt $p > /dev/null 2> $OUT/prof$p.err
  find $OUT/prof -name "p${p}_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt${p}_pf2.csv \;
  echo "== prompt $p"; grep -E "t16g|t16r|to_frag|rope_qk|flash|attention|attn" $OUT/kernel_stats_prompt${p}_pf2.csv | cut -c1-170
done
find $OUT -type f ! -name "*stats*" -size +2M -delete
