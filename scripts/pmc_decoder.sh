# rocprofv3 PMC pass over the stand-alone decoder driver (8 layers): SQ activity of every decode kernel
mkdir -p gpurun_out/mb
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_dec -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 > $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_dec.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_dec2 -o p -- $GRAFT_REPO_ROOT/build/bench_decoder 8 2048 >> $GRAFT_REPO_ROOT/gpurun_out/mb/pmc_dec.txt 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/mb -name "*.csv" -size +20M -delete
