#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 1500 python -m pytest tests -m gpu -v --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
grep -E "FAILED|PASSED|ERROR" gpurun_out/t_all.log | tail -5
grep -E "FAILED" gpurun_out/t_all.log | head
cat gpurun_out/summary.log
