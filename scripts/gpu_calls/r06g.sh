#!/bin/bash
# round 6, call 9: the short-prompt GEMM with activations shared through LDS (q4_gemm_t16g) and the rewritten fragment-order producer:
# op-level parity of every kernel / block shape, the model-level short-prompt cases with their tolerance statistics, timing at
# 17 / 64 / 128 / 256 tokens against the narrow kernel alone (EXL_GEMM_NO_T16G=1) and the op-by-op path, kernel stats at 128 / 256
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -40 > $OUT/tests_frag_ops.txt
cat $OUT/tests_frag_ops.txt
EXL_TOL_STATS=$OUT/tol_short.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -30 > $OUT/tests_short.txt
cat $OUT/tests_short.txt
EXL_GEMM_NO_T16G=1 EXL_TOL_STATS=$OUT/tol_short_t16r.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts and 7b" 2>&1 | tail -5 > $OUT/tests_short_t16r.txt
EXL_GEMM_NO_FRAG=1 EXL_TOL_STATS=$OUT/tol_short_opbyop.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts and 7b" 2>&1 | tail -5 > $OUT/tests_short_opbyop.txt
for p in 128 64 256 17; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" >> $OUT/short_prompt.txt
  EXL_GEMM_T16G_PF=0 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/LDS reads placed by the compiler (EXL_GEMM_T16G_PF=0): /' >> $OUT/short_prompt.txt
  EXL_GEMM_NO_T16G=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/narrow kernel only (EXL_GEMM_NO_T16G=1): /' >> $OUT/short_prompt.txt
  EXL_GEMM_NO_FRAG=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/op by op (EXL_GEMM_NO_FRAG=1): /' >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
for p in 128 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p$p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt $p > /dev/null 2> $OUT/prof$p.err
  find $OUT/prof -name "p${p}_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt$p.csv \;
  echo "== prompt $p"; grep -E "t16g|t16r|to_frag|rope_qk|flash|rms_norm|t16s|attention" $OUT/kernel_stats_prompt$p.csv | cut -c1-170
done
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
