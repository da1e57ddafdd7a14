#!/bin/bash
# round 5, call 16: wave-private activation image for the plain-vector ring launches (down_proj, o_proj at one KV split): parity of every
# ring class (ring == compiler stream bit for bit, real-shape executor, op-level fused ops) and the per-class times through bench_decoder
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT EXL_SKIP_SLOW=1
timeout 120 build/bench_decoder 32 2048 128 2 > $OUT/bench_decoder.txt 2>&1
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "ring_stream_equals or real_layer_shapes or executor_matches or stages_matches" > $OUT/t_model.log 2>&1; echo "rc $?" >> $OUT/t_model.log
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "q4_attn or q4_mlp or fused or dec_op" > $OUT/t_ops.log 2>&1; echo "rc $?" >> $OUT/t_ops.log
cat $OUT/bench_decoder.txt | cut -c1-220; tail -n 4 $OUT/t_model.log; tail -n 3 $OUT/t_ops.log
