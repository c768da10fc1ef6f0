#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
cd /tmp
for n in 4608 1152; do
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o eig$n -- python $GRAFT_REPO_ROOT/tools/eig_one.py $n > $GRAFT_REPO_ROOT/gpurun_out/eig_one_$n.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/gpurun_out/prof/eig${n}_results.db $GRAFT_REPO_ROOT/gpurun_out/stats_eig$n.md > /dev/null 2>&1
done
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof/*.db
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/eig_one_4608.log; head -12 gpurun_out/stats_eig4608.md | cut -c1-160; tail -1 gpurun_out/eig_one_1152.log; head -8 gpurun_out/stats_eig1152.md | cut -c1-160
