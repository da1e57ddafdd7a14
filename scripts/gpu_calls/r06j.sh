#!/bin/bash
# round 6, call 12: weights requested last in a step + vmcnt(6), slab pieces in the first half, RMSNorm sums of squares from the GEMM epilogues (to_frag spread over the chip)
# barrier per step): op-level parity, the model-level short-prompt cases on the GPTQ-statistics checkpoint, timing against the plain
# step (EXL_GEMM_T16G_PF=0) and the narrow kernel, SQ counters of both steps
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -k "frag" 2>&1 | tail -80 > $OUT/tests_frag_ops.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_frag_ops.txt
EXL_TOL_STATS=$OUT/tol_short.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -30 > $OUT/tests_short.txt
grep -E "^E  +Assert|passed|failed" $OUT/tests_short.txt
EXL_GEMM_NO_FRAG=1 EXL_TOL_STATS=$OUT/tol_short_opbyop.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q -k "short_prompts" 2>&1 | tail -5 > $OUT/tests_short_opbyop.txt
for p in 128 256 64 17; do
  timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" >> $OUT/short_prompt.txt
  EXL_GEMM_T16G_PF=0 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/plain step (EXL_GEMM_T16G_PF=0): /' >> $OUT/short_prompt.txt
  EXL_GEMM_NO_T16G=1 timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids" | sed 's/^/narrow kernel only (EXL_GEMM_NO_T16G=1): /' >> $OUT/short_prompt.txt
done
cat $OUT/short_prompt.txt
cd /tmp
for p in 128 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p$p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 8 --reps 5 --prompt $p > /dev/null 2> $OUT/prof$p.err
  find $OUT/prof -name "p${p}_kernel_stats*" -exec cp {} $OUT/kernel_stats_prompt$p.csv \;
  echo "== prompt $p"; grep -E "t16g|t16r|to_frag|rope_qk|flash|attention" $OUT/kernel_stats_prompt$p.csv | cut -c1-170
done
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 4 --reps 2 --prompt 128 > /dev/null 2> $OUT/pmc_$tag.err
done
python - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "t16" in n or "to_frag" in n and "retile" not in n:
            res[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(res.items()):
    print(n, {k: round(sum(x) / len(x), 1) for k, x in d.items()}, "launches", len(next(iter(d.values()))))
PY
find $OUT -name "*.csv" -size +2M -delete
find $OUT -type f ! -name "*stats*" -size +2M -delete
