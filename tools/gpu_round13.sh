#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
for i in 1 2 3; do
  AMD_LOG_LEVEL=1 timeout 900 python -X faulthandler -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_rep$i.log 2>&1
  echo "rep$i rc=$?" >> gpurun_out/summary.log
done
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
for i in 1 2 3; do tail -3 gpurun_out/t_rep$i.log | cut -c1-200; done
cat gpurun_out/summary.log
