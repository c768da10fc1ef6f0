#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "kernel tests rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py gram > gpurun_out/mb_gram.log 2>&1
echo "mb_gram rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py eig 576 2304 4608 > gpurun_out/mb_eig.log 2>&1
echo "mb_eig rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
tail -4 gpurun_out/t_kernels.log; tail -1 gpurun_out/bench.log | cut -c1-250; cat gpurun_out/summary.log
