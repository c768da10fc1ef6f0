#!/bin/bash
# first GPU session: kernel parity, backend parity, microbench
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/hw.log
nproc >> gpurun_out/hw.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_backend.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_backend.log 2>&1
echo "backend rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py gram > gpurun_out/mb_gram.log 2>&1
echo "mb_gram rc=$?" >> gpurun_out/summary.log
timeout 900 python tools/microbench.py eig 64 128 256 576 1152 > gpurun_out/mb_eig.log 2>&1
echo "mb_eig rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/t_kernels.log
tail -5 gpurun_out/t_backend.log
cat gpurun_out/summary.log
