mkdir -p gpurun_out; export TMPDIR=/tmp
CFG=${1:-12582914}
rm -rf gpurun_out/prof_steps
cd /tmp && LK_CONV_CONFIG=$CFG timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_steps -o p -- python $GRAFT_REPO_ROOT/tools/steps_only.py 16 > $GRAFT_REPO_ROOT/gpurun_out/prof_steps.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_steps -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r03_steps_only_kernel_stats_$CFG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_steps
head -16 gpurun_out/r03_steps_only_kernel_stats_$CFG.md | cut -c1-200
