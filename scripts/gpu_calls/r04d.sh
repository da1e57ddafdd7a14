#!/bin/bash
# round 4, call 4: tile order A/B (2-row rectangles per XCD vs the order of rounds 1-3) with its PMC traffic, the load-time fold of act-order
# down_proj (bit identity + 13B line), LoRA numbers
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_cold_launch_gpu.py -x -q -m gpu -k "gemm or dual or qkv or prompt_fusions or t16w or t16d2" > $OUT/t_ops.log 2>&1; echo "ops rc $?" >> $OUT/t_ops.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "folded or lora or end_to_end or golden or real_layer_shapes" > $OUT/t_model.log 2>&1; echo "model rc $?" >> $OUT/t_model.log
for mode in new old; do
  if [ $mode = old ]; then export EXL_GEMM_TILE_ROWS=0; else unset EXL_GEMM_TILE_ROWS; fi
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gen 16 > $OUT/bench_7b_$mode.json 2> $OUT/bench_7b_$mode.err
  timeout 500 python bench.py --model 13b --act-order --steps 2 --warmup 1 --no-cpu-baseline --gen 16 > $OUT/bench_13b_act_$mode.json 2> $OUT/bench_13b_act_$mode.err
done
unset EXL_GEMM_TILE_ROWS
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_new -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --layers 4 --reps 2 > /dev/null 2> $OUT/pmc_new.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof13 -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_once.py --model 13b --act-order gptq --layers 4 --reps 3 > /dev/null 2> $OUT/prof13.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
res = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_new/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and ("q4_gemm" in r["Kernel_Name"] or "flash" in r["Kernel_Name"]):
            res[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in res.items():
    print("FETCH_SIZE x2 MB", k, round(2 * sum(v) / len(v) / 1024, 1), len(v))
for f in glob.glob("$OUT/prof13/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 12: print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
find $OUT -name "*.csv" -size +2M -delete
timeout 600 python scripts/bench_lora.py --out $OUT/lora.json > $OUT/lora.log 2>&1
tail -n 3 $OUT/t_ops.log; tail -n 3 $OUT/t_model.log; tail -n 2 $OUT/lora.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "prefill", d.get("prefill_tokens_per_s"), d.get("prefill_ms"), "decode", d["value"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
