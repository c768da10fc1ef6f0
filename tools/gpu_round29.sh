#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
tail -3 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log | cut -c1-1500; cat gpurun_out/summary.log
