#!/bin/bash
# tensor-parallel code path on ONE GPU: (a) world 1 without a process group, (b) with a one-rank RCCL group and the collectives
# forced, captured in the per-token hipGraph; the normal single-GPU line beside them
mkdir -p gpurun_out
timeout 600 python bench.py --tensor-parallel --no-cpu-baseline > gpurun_out/r02y_tp1.json 2> gpurun_out/r02y_tp1.err
EXL_BENCH_FORCE_DIST=1 EXL_TP_ALWAYS_COLLECTIVE=1 MASTER_PORT=29533 timeout 600 python bench.py --tensor-parallel --no-cpu-baseline > gpurun_out/r02y_tp1_rccl.json 2> gpurun_out/r02y_tp1_rccl.err
EXL_BENCH_FORCE_DIST=1 EXL_TP_ALWAYS_COLLECTIVE=1 MASTER_PORT=29534 timeout 600 python bench.py --tensor-parallel --no-cpu-baseline --no-graph > gpurun_out/r02y_tp1_rccl_eager.json 2> gpurun_out/r02y_tp1_rccl_eager.err
tail -c 900 gpurun_out/r02y_tp1.json; tail -3 gpurun_out/r02y_tp1.err
tail -c 900 gpurun_out/r02y_tp1_rccl.json; tail -5 gpurun_out/r02y_tp1_rccl.err
tail -c 900 gpurun_out/r02y_tp1_rccl_eager.json; tail -3 gpurun_out/r02y_tp1_rccl_eager.err
