#!/bin/bash
# round 5, call 12: the same with one-rank RCCL groups everywhere (EXL_BENCH_FORCE_DIST=1: process group "nccl" with one rank; the hop of
# the layer split captured in the graph, the tensor-parallel collectives forced and captured)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
EXL_BENCH_FORCE_DIST=1 EXL_TP_ALWAYS_COLLECTIVE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline-probe --sharded-at-one-gpu > $OUT/bench_sharded_1gpu_rccl.json 2> $OUT/bench_sharded_1gpu_rccl.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_sharded_1gpu_rccl.json").read().strip().splitlines()[-1])
    print("headline", d["value"], d["prefill_tokens_per_s"])
    for k, v in (d.get("sharded") or {}).items():
        print(k, v.get("value"), v.get("prefill_tokens_per_s"), v.get("rccl_ranks"), v.get("backend"), v.get("logits_finite"), v.get("seconds"), (v.get("decode_mode") or "")[-80:], v.get("error"))
except Exception as e:
    print("ERR", e)
PY
tail -n 4 $OUT/bench_sharded_1gpu_rccl.err | cut -c1-300
