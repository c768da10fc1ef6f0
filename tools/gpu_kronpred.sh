mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/kron_predictive_c4.py > gpurun_out/kronpred.log 2>&1
echo "kronpred rc=$?" > gpurun_out/summary_kp.log
rm -rf gpurun_out/prof_kp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kp -o kp -- python $GRAFT_REPO_ROOT/tools/kron_predictive_c4.py --profile > $GRAFT_REPO_ROOT/gpurun_out/kronpred_prof.log 2>&1
echo "rocprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary_kp.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_kp/kp_results.db gpurun_out/stats_kronpred.md > /dev/null 2>&1
rm -rf gpurun_out/prof_kp
tail -2 gpurun_out/kronpred.log; head -25 gpurun_out/stats_kronpred.md; cat gpurun_out/summary_kp.log
