mkdir -p gpurun_out/mb
./build/bench_gemm 2048 20 > gpurun_out/mb/gemm1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_gemm -o p -- $GRAFT_REPO_ROOT/build/bench_gemm 2048 4 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_gemm2 -o p -- $GRAFT_REPO_ROOT/build/bench_gemm 2048 4 > /dev/null 2>&1
