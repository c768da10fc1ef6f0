export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() {  # tag, counters...
  tag=$1; shift
  rm -rf $R/gpurun_out/pmcw
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmcw -o c -- python $R/tools/winp_pmc_run.py > $R/gpurun_out/pmcw_$tag.log 2>&1)
  python $R/tools/rocpd_pmc.py $R/gpurun_out/r04_pmc_winp_$tag.md $(find $R/gpurun_out/pmcw -name "*.db") > /dev/null 2>&1
  rm -rf $R/gpurun_out/pmcw
  grep "conv_\|kernel" $R/gpurun_out/r04_pmc_winp_$tag.md | cut -c1-400
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass b SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass c TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass d FETCH_SIZE
pass e WRITE_SIZE
