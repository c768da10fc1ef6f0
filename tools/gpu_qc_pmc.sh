# PMC pass over the weight-sharing predictive kernel on the c4 layer shapes (tools/quadconv_bench.py) + refreshed
# end-to-end numbers of the Kron predictive and the exact diagonal on ResNet-18.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_qc
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_qc -o qc -- python $GRAFT_REPO_ROOT/tools/quadconv_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_qc.log 2>&1
echo "pmc rc=$?" > $GRAFT_REPO_ROOT/gpurun_out/summary_qcp.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py gpurun_out/pmc_quadconv.md gpurun_out/pmc_qc/qc_results.db >> gpurun_out/summary_qcp.log 2>&1
rm -rf gpurun_out/pmc_qc
timeout 300 python -m pytest tests/test_weight_sharing.py tests/test_gpu_backend.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_ws.log 2>&1
echo "tests rc=$?" >> gpurun_out/summary_qcp.log
timeout 600 python tools/kron_predictive_c4.py > gpurun_out/kronpred.log 2>&1
echo "kronpred rc=$?" >> gpurun_out/summary_qcp.log
timeout 600 python tools/diag_c4.py > gpurun_out/diag_c4.log 2>&1
echo "diag rc=$?" >> gpurun_out/summary_qcp.log
cat gpurun_out/pmc_quadconv.md | cut -c1-400; tail -2 gpurun_out/t_ws.log; tail -1 gpurun_out/kronpred.log; tail -1 gpurun_out/diag_c4.log; cat gpurun_out/summary_qcp.log
