#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py gram > gpurun_out/mb_gram.log 2>&1
echo "mb_gram rc=$?" >> gpurun_out/summary.log
timeout 900 python tools/eig_debug.py 2 > gpurun_out/eig_debug.log 2>&1
echo "eig_debug rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
tail -4 gpurun_out/t_all.log; tail -2 gpurun_out/bench.log; cat gpurun_out/summary.log
