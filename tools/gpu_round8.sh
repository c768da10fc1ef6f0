#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 600 python tools/race_debug.py > gpurun_out/race_debug.log 2>&1
echo "race_debug rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predictive > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
grep -v amdgpu gpurun_out/race_debug.log; tail -1 gpurun_out/bench.log | cut -c1-300; cat gpurun_out/summary.log
