#!/bin/bash
# round 6, call 14: o_proj / down_proj with K cut over blocks (the last block at a tile adds the slices): parity (op level, model
# level), timing for 2 / 4 / 8 ranges and the launcher's choice against K whole, kernel stats
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -60 > $OUT/tests_frag_ops.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_frag_ops.txt
EXL_TOL_STATS=$OUT/tol_short.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -30 > $OUT/tests_short.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_short.txt
for p in 128 256 64 17; do
  for ks in auto 0 2 4 8; do
    if [ $ks = auto ]; then unset EXL_GEMM_KSPLIT; else export EXL_GEMM_KSPLIT=$ks; fi
    timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/K ranges $ks: /" >> $OUT/short_prompt.txt
  done
done
unset EXL_GEMM_KSPLIT
cat $OUT/short_prompt.txt
cd /tmp
for p in 128 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p$p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt $p > /dev/null 2> $OUT/prof$p.err
  find $OUT/prof -name "p${p}_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt$p.csv \;
  echo "== prompt $p"; grep -E "t16g|t16r|to_frag|rope_qk|flash|attention" $OUT/kernel_stats_prompt$p.csv | cut -c1-170
done
find $OUT -type f ! -name "*stats*" -size +2M -delete
