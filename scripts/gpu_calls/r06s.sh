#!/bin/bash
# round 6, call 23: round-end profiles (rocprofv3 kernel stats of the bench command, PMC traffic) at git head 44e7552, then the driver-style bench line
set -u
export GIT_HEAD=44e7552
cd $GRAFT_REPO_ROOT
bash scripts/gpu_r06_profiles.sh 2>&1 | tail -40
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06s
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "prefill_tokens_per_s", "decode_best_tokens_per_s")})
print(d.get("other_lengths"))
r = d["roofline"]; print({k: r.get(k) for k in ("frac", "avg_launch_us", "rocprof_avg_us", "rocprof_source", "token_ms_sum_of_classes", "traffic", "traffic_source")})
x = d.get("dropin_reference_model_py"); print(x and {k: x.get(k) for k in ("prefill_tokens_per_s", "decode_worst_tokens_per_s", "decode_best_tokens_per_s")})
print({k: (v.get("value"), v.get("prefill_tokens_per_s")) for k, v in d.get("other_configs", {}).items()})
PY
