mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_sweep_nhwc.py tests/test_gpu_timed_config.py tests/test_gpu_dynamic_range.py tests/test_gpu_baseline_parity.py -m gpu -x -q > gpurun_out/r04_winp_tests.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/r04_winp_tests.log
