#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log gpurun_out/prof
timeout 600 python -m pytest tests/test_gpu_backend.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "sweep or kfac or vjp" > gpurun_out/t_sweep.log 2>&1
echo "sweep tests rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-predictive --no-eigh > gpurun_out/bench_sweep.log 2>&1
echo "bench sweep rc=$?" >> gpurun_out/summary.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r5 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof/r5_results.db gpurun_out/stats_r5.md > /dev/null 2>&1
rm -f gpurun_out/prof/*.db
tail -3 gpurun_out/t_sweep.log; tail -1 gpurun_out/bench_sweep.log | cut -c1-300; cat gpurun_out/summary.log
