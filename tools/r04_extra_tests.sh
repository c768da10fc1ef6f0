mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -x -q -k "benched or syevj or eigendecomposition" > gpurun_out/r04_extra_tests.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r04_extra_tests.log; sort -g -r gpurun_out/parity_errors.log | grep -v "^#" | head -40
