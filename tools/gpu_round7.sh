#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 600 python tools/fused_debug.py > gpurun_out/fused_debug.log 2>&1
echo "fused_debug rc=$?" >> gpurun_out/summary.log
timeout 900 python tools/microbench.py eig 576 1152 2304 4608 > gpurun_out/mb_eig.log 2>&1
echo "mb_eig rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/eig_debug.py 2 > gpurun_out/eig_debug.log 2>&1
echo "eig_debug rc=$?" >> gpurun_out/summary.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o e1 -- python $GRAFT_REPO_ROOT/tools/microbench.py eig 4608 > $GRAFT_REPO_ROOT/gpurun_out/prof_eig.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof/e1_results.db gpurun_out/stats_eig4608.md > /dev/null 2>&1
rm -f gpurun_out/prof/*.db
cat gpurun_out/summary.log
