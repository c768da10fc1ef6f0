#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/microbench.py gram > gpurun_out/mb_gram.log 2>&1
echo "mb_gram rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof/r2_results.db gpurun_out/stats_r2.md > /dev/null 2>&1
rm -f gpurun_out/prof/*.db
tail -4 gpurun_out/t_all.log; tail -2 gpurun_out/bench.log; cat gpurun_out/summary.log
