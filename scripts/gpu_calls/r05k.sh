#!/bin/bash
# round 5, call 11: the N > 1 machinery of bench.py on the one GPU -- the sharded sub-runs launched by rank 0 as torch.distributed.run jobs
# (one-rank RCCL groups: layer split with the hop in the graph, tensor parallel with the collectives forced), behind a --gpus 1 headline
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
EXL_TP_ALWAYS_COLLECTIVE=1 timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline-probe --sharded-at-one-gpu > $OUT/bench_sharded_1gpu.json 2> $OUT/bench_sharded_1gpu.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_sharded_1gpu.json").read().strip().splitlines()[-1])
    print("headline", d["value"], d["prefill_tokens_per_s"])
    for k, v in (d.get("sharded") or {}).items():
        print(k, v.get("value"), v.get("prefill_tokens_per_s"), v.get("rccl_ranks"), v.get("backend"), v.get("logits_finite"), v.get("seconds"), (v.get("decode_mode") or "")[:90], v.get("error"))
except Exception as e:
    print("ERR", e)
PY
tail -n 4 $OUT/bench_sharded_1gpu.err | cut -c1-300
