mkdir -p gpurun_out
python tools/kron_predictive_c4.py --profile 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/exp_r03_r.log
timeout 600 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "predictive or c2 or c4" 2>&1 | tail -3 | tee -a gpurun_out/exp_r03_r.log
