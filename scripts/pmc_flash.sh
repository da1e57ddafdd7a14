mkdir -p gpurun_out/pmc_flash
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
EXL_FLASH_4WAVE=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_flash/a -o p -- python $GRAFT_REPO_ROOT/scripts/flash_check.py > /dev/null 2>&1
EXL_FLASH_4WAVE=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_flash/b -o p -- python $GRAFT_REPO_ROOT/scripts/flash_check.py > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/pmc_flash -type f -size +6M -delete
