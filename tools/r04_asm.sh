export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_timed_config.py tests/test_gpu_backend.py -m gpu -x -q -k "pixpair or assemble or eleven or R4 or kfac" 2>&1 | tail -3
python tools/finalize_host.py 20 2>&1 | grep "finalize:"
python tools/assemble_bench.py 2>&1 | grep -v amdgpu.ids | tail -8
bash tools/r04_k20.sh
