# patch form in-step (non-fused / unfused launches) + a clean steady-state kernel trace of the step
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_a.log; : > $O
for cfg in "LK_CONV_CONFIG=2" "LK_CONV_CONFIG=0" "LK_CONV_CONFIG=2 LK_FUSE_VJP=0" "LK_CONV_CONFIG=0 LK_FUSE_VJP=0"; do
  for rep in 1 2; do echo "$cfg: $(env $cfg python tools/steps_only.py 48 2>&1 | tail -1)" >> $O; done
done
rm -rf gpurun_out/prof_steps
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_steps -o p -- python $GRAFT_REPO_ROOT/tools/steps_only.py 16 > $GRAFT_REPO_ROOT/gpurun_out/prof_steps.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_steps -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r03_steps_only_kernel_stats_a.md > /dev/null 2>&1
rm -rf gpurun_out/prof_steps
cat $O; head -60 gpurun_out/r03_steps_only_kernel_stats_a.md
