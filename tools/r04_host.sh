export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv.py tests/test_gpu_kron_algebra.py tests/test_gpu_timed_config.py -m gpu -x -q -n 2 2>&1 | tail -2
python tools/sustained.py 4 100 2>&1 | grep chunk
bash tools/r04_k20.sh
