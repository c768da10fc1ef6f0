mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_gf
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gf -o p -- python $GRAFT_REPO_ROOT/tools/gram_fuse_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_gf.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_gf -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/exp_r03_d.md > /dev/null 2>&1
rm -rf gpurun_out/prof_gf
tail -4 gpurun_out/prof_gf.log; head -14 gpurun_out/exp_r03_d.md | cut -c1-220
