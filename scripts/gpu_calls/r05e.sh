#!/bin/bash
# round 5, call 5: the HIP half of the full-depth perplexity records, the data dependence of the prompt rate (uniform vs centered nibbles,
# same box, alternating), the whole GPU suite, and the round's profiles
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT EXL_SKIP_SLOW=1
timeout 400 python scripts/ppl_full_depth.py --model 7b --seeds 17,18 --hip-only $OUT/ppl > $OUT/ppl_7b.log 2>&1
timeout 500 python scripts/ppl_full_depth.py --model 13b --act-order --seeds 17 --hip-only $OUT/ppl > $OUT/ppl_13b.log 2>&1
for rep in 1 2; do
  for nib in uniform centered; do
    timeout 300 python bench.py --brief --no-roofline-probe --steps 4 --warmup 2 --nibbles $nib > $OUT/ab_7b_${nib}_$rep.json 2>/dev/null
    timeout 400 python bench.py --brief --no-roofline-probe --model 65b --steps 2 --warmup 1 --nibbles $nib > $OUT/ab_65b_${nib}_$rep.json 2>/dev/null
  done
done
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 1700 python -m pytest tests -q -m gpu > $OUT/full_tests.log 2>&1; echo "suite rc $?" >> $OUT/full_tests.log
GIT_HEAD=$(cat $GRAFT_REPO_ROOT/.git_head 2>/dev/null || echo unknown) bash scripts/gpu_r05_profiles.sh > $OUT/profiles.log 2>&1
tail -n 2 $OUT/ppl_7b.log $OUT/ppl_13b.log | cut -c1-400
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "decode", d["value"], "prefill", d["prefill_tokens_per_s"], "finite", d["logits_finite"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -n 6 $OUT/full_tests.log
tail -n 30 $OUT/profiles.log | cut -c1-200
