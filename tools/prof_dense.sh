cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_dense -- python $R/tools/profile_dense.py > $R/gpurun_out/pmc_dense.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_dense2 -- python $R/tools/profile_dense.py >> $R/gpurun_out/pmc_dense.log 2>&1
ls -R $R/gpurun_out/pmc_dense | head
