# Secondary measurements with the final kernels: small configs, c5 (BERT-base), prior sweep on the c4 posterior, PMC
# pass over the weight-sharing predictive kernel.  Writes under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/small_configs.py > gpurun_out/small.log 2>&1; echo "small rc=$?" > gpurun_out/summary_ev.log
timeout 600 python tools/c5_bert.py > gpurun_out/c5.log 2>&1; echo "c5 rc=$?" >> gpurun_out/summary_ev.log
timeout 300 python tools/marglik_bench.py > gpurun_out/marglik.log 2>&1; echo "marglik rc=$?" >> gpurun_out/summary_ev.log
rm -rf gpurun_out/pmc_qc
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_qc -o qc -- python $GRAFT_REPO_ROOT/tools/quadconv_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_qc.log 2>&1
echo "pmc rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary_ev.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py gpurun_out/pmc_quadconv.md gpurun_out/pmc_qc/qc_results.db >> gpurun_out/summary_ev.log 2>&1
rm -rf gpurun_out/pmc_qc
tail -1 gpurun_out/small.log | cut -c1-600; tail -1 gpurun_out/c5.log | cut -c1-700; tail -1 gpurun_out/marglik.log; cat gpurun_out/pmc_quadconv.md | cut -c1-330; cat gpurun_out/summary_ev.log
