# PMC pass over the split-fp16 convolution on the c4 sweep shapes (tools/conv_f16x2_bench.py): matrix-pipe busy, wait
# breakdown, LDS bank conflicts.  usage: pmc_conv.sh tag
TAG=${1:-x}
export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_conv
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_conv -o c -- python $GRAFT_REPO_ROOT/tools/conv_f16x2_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_conv.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py gpurun_out/pmc_conv_$TAG.md $(find gpurun_out/pmc_conv -name "*.db") > /dev/null 2>&1
rm -rf gpurun_out/pmc_conv
cat gpurun_out/pmc_conv_$TAG.md | cut -c1-300
