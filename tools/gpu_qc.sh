mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "shared" > gpurun_out/t_qc.log 2>&1
echo "tests rc=$?" > gpurun_out/summary_qc.log
LK_QC_B6=1 timeout 300 python tools/quadconv_bench.py > gpurun_out/qc_v4.log 2>&1
LK_QC_B6=0 timeout 300 python tools/quadconv_bench.py > gpurun_out/qc_v1.log 2>&1
tail -2 gpurun_out/t_qc.log; echo B6; grep -v amdgpu.ids gpurun_out/qc_v4.log | cut -c1-120; echo FP32; grep -v amdgpu.ids gpurun_out/qc_v1.log | cut -c1-120
