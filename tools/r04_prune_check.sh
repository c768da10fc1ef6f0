mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_sweep_nhwc.py tests/test_gpu_switches.py tests/test_gpu_timed_config.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r04_prune_tests.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/r04_prune_tests.log
for rep in 1 2; do echo "step: $(timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)"; done
