#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 900 python tools/microbench.py eig 576 1152 2304 4608 > gpurun_out/mb_eig.log 2>&1
echo "mb_eig rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive --no-eigh --no-overlap > gpurun_out/bench_noovl.log 2>&1
echo "bench-noovl rc=$?" >> gpurun_out/summary.log
tail -6 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log; tail -1 gpurun_out/bench_noovl.log; cat gpurun_out/summary.log
