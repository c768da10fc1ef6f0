mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_weight_sharing.py -m gpu -q --tb=short -p no:cacheprovider -k "shared or sequence" > gpurun_out/t_dg.log 2>&1
echo "tests rc=$?" > gpurun_out/summary_dg.log
timeout 600 python -m pytest tests/test_gpu_backend.py -m gpu -q --tb=short -p no:cacheprovider -k "diag" > gpurun_out/t_dg2.log 2>&1
echo "backend diag tests rc=$?" >> gpurun_out/summary_dg.log
timeout 600 python tools/diag_c4.py > gpurun_out/diag_c4.log 2>&1
echo "diag_c4 rc=$?" >> gpurun_out/summary_dg.log
tail -2 gpurun_out/t_dg.log; tail -2 gpurun_out/t_dg2.log; tail -1 gpurun_out/diag_c4.log; cat gpurun_out/summary_dg.log
