#!/bin/bash
# round 5, call 7: prompt-pass switches re-measured on FINITE data (rounds 3-4 decided them on inf / NaN or |x| ~ 4e4 activations):
# tile order per XCD (EXL_GEMM_TILE_ROWS=2), K-split of partly filled last rounds, alternating on one box
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in default tile2 notail; do
    case $cfg in default) ENV="";; tile2) ENV="EXL_GEMM_TILE_ROWS=2";; notail) ENV="EXL_GEMM_NO_TAIL_SPLIT=1";; esac
    env $ENV timeout 300 python bench.py --brief --no-roofline-probe --steps 4 --warmup 2 > $OUT/7b_${cfg}_$rep.json 2>/dev/null
    env $ENV timeout 300 python bench.py --brief --no-roofline-probe --model 13b --act-order --steps 3 --warmup 1 > $OUT/13bact_${cfg}_$rep.json 2>/dev/null
    env $ENV timeout 400 python bench.py --brief --no-roofline-probe --model 65b --steps 2 --warmup 1 > $OUT/65b_${cfg}_$rep.json 2>/dev/null
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "prefill", d["prefill_tokens_per_s"], "decode", d["value"], "finite", d["logits_finite"])
    except Exception as e:
        print(f, "ERR", e)
PY
