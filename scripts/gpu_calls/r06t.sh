#!/bin/bash
# round 6, call 24: the whole GPU suite (as the driver runs it) with its tolerance statistics, smoke, the bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06t
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
EXL_TOL_STATS=$OUT/model_tolerance_stats.jsonl timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -30 > $OUT/tests_gpu.txt
tail -8 $OUT/tests_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for p in 16 128; do timeout 200 python scripts/prefill_once.py --layers 8 --reps 5 --prompt $p --time 2>&1 | grep -v "^ok\|amdgpu.ids"; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json | head -c 600; echo
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "vs_baseline")})
print(d.get("other_lengths"))
print(d.get("batched_decode"))
PY
