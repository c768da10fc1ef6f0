mkdir -p gpurun_out; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
P2="SQ_WAVES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH"
rm -rf gpurun_out/pmc_win
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/gpurun_out/pmc_win/p$i -o c -- python $GRAFT_REPO_ROOT/tools/win_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_win_$i.log 2>&1
  cd $GRAFT_REPO_ROOT
done
python tools/rocpd_pmc.py gpurun_out/r03_pmc_win_b.md $(find gpurun_out/pmc_win -name "*.db") > /dev/null 2>&1
rm -rf gpurun_out/pmc_win
grep -E "conv_|kernel \|" gpurun_out/r03_pmc_win_b.md | cut -c1-900
