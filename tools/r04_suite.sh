mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-v2}
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=10 > gpurun_out/r04_gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r04_gpu_tests_$TAG.log
cp gpurun_out/parity_errors.log gpurun_out/r04_parity_errors_$TAG.log
