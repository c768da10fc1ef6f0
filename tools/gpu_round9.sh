#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 900 python tools/race_debug2.py > gpurun_out/race_debug2.log 2>&1
echo "race_debug2 rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py gram > gpurun_out/mb_gram.log 2>&1
echo "mb_gram rc=$?" >> gpurun_out/summary.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/summary.log
grep -v amdgpu gpurun_out/race_debug2.log; tail -3 gpurun_out/t_kernels.log; cat gpurun_out/summary.log
