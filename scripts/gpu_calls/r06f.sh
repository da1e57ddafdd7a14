#!/bin/bash
# round 6, call 8: first run of the short-prompt layer path (exl_q4_layer_prompt: GEMMs on fragment-order activations): parity at real
# shapes, the model tests whose short prompts now take it, timing at 64 / 128 / 256 tokens; and the text of the 13B act-order full-depth
# perplexity case for its golden file
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "short_prompts" 2>&1 | tail -15 > $OUT/tests_short.txt
cat $OUT/tests_short.txt
for p in 128 64 256 17; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" >> $OUT/short_prompt.txt
  EXL_GEMM_NO_FRAG=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/op by op (EXL_GEMM_NO_FRAG=1): /' >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p128 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 128 > /dev/null 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt128.csv \;
find $OUT/prof -type f ! -name "*stats*" -size +2M -delete
grep -E "t16r|to_frag|rope_qk|flash|rms_norm|t16s" $OUT/kernel_stats_prompt128.csv | cut -c1-170
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "not full_depth and not short_prompts" 2>&1 | tail -8 > $OUT/tests_model.txt
cat $OUT/tests_model.txt
timeout 600 python scripts/ppl_full_depth.py --model 13b --act-order --seeds 17 --hip-only $OUT/ppl > $OUT/ppl13.log 2>&1
tail -2 $OUT/ppl13.log | cut -c1-400
