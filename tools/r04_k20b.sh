export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_timed_config.py tests/test_gpu_switches.py tests/test_gpu_backend.py -m gpu -x -q > gpurun_out/r04_k20b_tests.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/r04_k20b_tests.log
bash tools/r04_k20.sh
echo "step: $(timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)"
