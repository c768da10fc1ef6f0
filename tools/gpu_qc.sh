mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "quadform_shared" > gpurun_out/t_qc.log 2>&1
echo "tests rc=$?" > gpurun_out/summary_qc.log
LK_QC_PK=1 timeout 300 python tools/quadconv_bench.py > gpurun_out/qc_pk1.log 2>&1
LK_QC_PK=0 timeout 300 python tools/quadconv_bench.py > gpurun_out/qc_pk0.log 2>&1
tail -2 gpurun_out/t_qc.log; echo PK1; cat gpurun_out/qc_pk1.log; echo PK0; cat gpurun_out/qc_pk0.log
