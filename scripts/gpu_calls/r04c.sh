#!/bin/bash
# round 4, call 3: the whole GPU suite (sampler big path, head GEMM, token ring, tail split, binding ...), layer-split bench on one rank, profiles
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
EXL_TOL_STATS=$OUT/tol_stats.jsonl timeout 1500 python -m pytest tests -q -m gpu > $OUT/full_tests.log 2>&1; echo "suite rc $?" >> $OUT/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
timeout 400 python bench.py --layer-split --gpus 1 --steps 2 --warmup 1 > $OUT/bench_layer_split_1rank.json 2> $OUT/bench_layer_split_1rank.err
EXL_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --layer-split --gpus 1 --steps 2 --warmup 1 > $OUT/bench_layer_split_1rank_rccl.json 2> $OUT/bench_layer_split_1rank_rccl.err
bash scripts/gpu_r04_profiles.sh > $OUT/profiles.log 2>&1
tail -n 6 $OUT/full_tests.log; tail -n 2 $OUT/smoke.log; tail -c 700 $OUT/bench_layer_split_1rank.json; tail -c 400 $OUT/bench_layer_split_1rank_rccl.json; tail -n 20 $OUT/profiles.log
