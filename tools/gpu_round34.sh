#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 600 python -m pytest tests/test_gpu_backend.py -m gpu -q --tb=short -p no:cacheprovider -k "sweep or kfac" > gpurun_out/t_sweep.log 2>&1
echo "sweep tests rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-predictive --no-eigh > gpurun_out/bench_a.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
timeout 300 python tools/torch_prof_step.py 2>&1 | grep "^COPY" | wc -l >> gpurun_out/summary.log
tail -2 gpurun_out/t_sweep.log; tail -1 gpurun_out/bench_a.log | cut -c1-260; cat gpurun_out/summary.log
