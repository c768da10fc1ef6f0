#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 600 python -m pytest tests/test_mc_fisher.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_mc.log 2>&1
echo "mc tests rc=$?" >> gpurun_out/summary.log
timeout 600 python tools/small_configs.py > gpurun_out/small.log 2>&1
echo "small rc=$?" >> gpurun_out/summary.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o small -- python $GRAFT_REPO_ROOT/tools/small_configs.py > $GRAFT_REPO_ROOT/gpurun_out/prof_small.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof/small_results.db gpurun_out/stats_small.md > /dev/null 2>&1
rm -f gpurun_out/prof/*.db
tail -3 gpurun_out/t_mc.log; grep -v "Cannot find" gpurun_out/small.log | tail -8; cat gpurun_out/summary.log
