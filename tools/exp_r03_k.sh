mkdir -p gpurun_out; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
P2="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
rm -rf gpurun_out/pmc_win
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/gpurun_out/pmc_win/p$i -o c -- python $GRAFT_REPO_ROOT/tools/win_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_win_$i.log 2>&1
  cd $GRAFT_REPO_ROOT
done
python tools/rocpd_pmc.py gpurun_out/r03_pmc_win_a.md $(find gpurun_out/pmc_win -name "*.db") > /dev/null 2>&1
rm -rf gpurun_out/pmc_win
grep -E "conv_|kernel \|" gpurun_out/r03_pmc_win_a.md | cut -c1-1200
