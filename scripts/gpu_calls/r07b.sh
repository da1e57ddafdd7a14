#!/bin/bash
# round 6, call 38 (round end, again): profiles at git head 79ff7e9 (rocprofv3 kernel stats of the bench command, PMC traffic), the whole GPU suite, smoke, the bench line
set -u
export GIT_HEAD=79ff7e9
cd $GRAFT_REPO_ROOT
bash scripts/gpu_r06_profiles.sh > /dev/null 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r07b
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/model_tolerance_stats.jsonl timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -8 > $OUT/tests_gpu.txt
tail -4 $OUT/tests_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "prefill_tokens_per_s", "decode_best_tokens_per_s")})
print({k: v.get("prefill_ms") for k, v in d.get("other_lengths", {}).items()})
r = d["roofline"]; print({k: r.get(k) for k in ("frac", "avg_launch_us", "rocprof_avg_us", "rocprof_source", "token_ms_sum_of_classes")}, r.get("traffic_source", "")[-80:])
x = d.get("dropin_reference_model_py"); print(x and {k: x.get(k) for k in ("prefill_tokens_per_s", "decode_worst_tokens_per_s", "decode_best_tokens_per_s")})
print({k: (v.get("value"), v.get("prefill_tokens_per_s")) for k, v in d.get("other_configs", {}).items()})
print(d.get("batched_decode"))
PY
