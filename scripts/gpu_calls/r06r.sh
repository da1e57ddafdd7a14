#!/bin/bash
# round 6, call 22: K cut over blocks again, the last arrival reading four slices per round trip
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -5
for p in 128 100 64; do
  for ks in 0 -1 2 4 8; do
    EXL_GEMM_KSPLIT=$ks timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed "s/^/K ranges $ks: /" >> $OUT/short_prompt.txt
  done
done
cat $OUT/short_prompt.txt
cd /tmp
EXL_GEMM_KSPLIT=-1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p128 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 128 > /dev/null 2> $OUT/prof128.err
find $OUT/prof -name "p128_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt128_kcut.csv \;
grep -E "t16g|t16r" $OUT/kernel_stats_prompt128_kcut.csv | cut -c1-170
find $OUT -type f ! -name "*stats*" -size +2M -delete
