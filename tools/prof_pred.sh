# rocprofv3 kernel trace of the c4 Kron GLM predictive (tools/kron_predictive_c4.py --profile) -> gpurun_out/prof_pred_<tag>.md
TAG=${1:-x}
export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_pred_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pred_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/kron_predictive_c4.py --profile > $GRAFT_REPO_ROOT/gpurun_out/prof_pred_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_pred_$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/prof_pred_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_pred_$TAG
head -32 gpurun_out/prof_pred_$TAG.md; tail -2 gpurun_out/prof_pred_$TAG.log
