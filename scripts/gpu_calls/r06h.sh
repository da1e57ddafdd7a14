#!/bin/bash
# round 6, call 10: staggered walk over K (blocks of an XCD one step apart: does the L2 stop fetching every activation line once per
# block?): op-level parity of the fragment-order GEMMs, model-level cases (13B on all three paths), timing with / without the stagger,
# L2 counters of both
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -40 > $OUT/tests_frag_ops.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_frag_ops.txt
EXL_TOL_STATS=$OUT/tol_short.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -30 > $OUT/tests_short.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_short.txt
EXL_GEMM_NO_T16G=1 EXL_TOL_STATS=$OUT/tol_short_t16r.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -5 > $OUT/tests_short_t16r.txt
EXL_GEMM_NO_FRAG=1 EXL_TOL_STATS=$OUT/tol_short_opbyop.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -5 > $OUT/tests_short_opbyop.txt
for p in 128 256 64; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" >> $OUT/short_prompt.txt
  EXL_GEMM_NO_STAGGER=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/every block starts at row-block 0 (EXL_GEMM_NO_STAGGER=1): /' >> $OUT/short_prompt.txt
  EXL_GEMM_NO_T16G=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/narrow kernel only (EXL_GEMM_NO_T16G=1): /' >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
for v in stagger nostagger; do
  if [ $v = nostagger ]; then export EXL_GEMM_NO_STAGGER=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p128 -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt 128 > /dev/null 2> $OUT/prof_$v.err
  find $OUT/prof_$v -name "p128_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt128_$v.csv \;
  echo "== $v"; grep -E "t16g|t16r|to_frag|rope_qk|flash|attention" $OUT/kernel_stats_prompt128_$v.csv | cut -c1-170
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${v}_$tag -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 4 --reps 2 --prompt 128 > /dev/null 2> $OUT/pmc_${v}_$tag.err
  done
done
unset EXL_GEMM_NO_STAGGER
python - <<PY
import csv, glob, collections
for v in ("stagger", "nostagger"):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/pmc_%s_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "t16" in n or "to_frag" in n:
                res[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", v)
    for n, d in sorted(res.items()):
        print(n, {k: round(sum(x) / len(x), 1) for k, x in d.items()}, "launches", len(next(iter(d.values()))))
PY
find $OUT -name "*.csv" -size +2M -delete
find $OUT -type f ! -name "*stats*" -size +2M -delete
