#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log gpurun_out/pmc
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py gram > gpurun_out/mb_gram.log 2>&1
echo "mb_gram rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py eig 576 2304 4608 > gpurun_out/mb_eig.log 2>&1
echo "mb_eig rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log | cut -c1-300; cat gpurun_out/summary.log
