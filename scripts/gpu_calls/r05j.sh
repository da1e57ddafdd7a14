#!/bin/bash
# round 5, call 10: op-level cases for head_dim 100 / 36 / 200 (RoPE pair kernel, 4-byte cache copies, 8-byte attention accesses)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "rope or update_cache or attention_decode" > $OUT/t.log 2>&1; echo "rc $?" >> $OUT/t.log
tail -n 15 $OUT/t.log
