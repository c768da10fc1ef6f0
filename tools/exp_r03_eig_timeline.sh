mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_eig
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_eig -o p -- python $GRAFT_REPO_ROOT/tools/eig_profile.py decompose > /tmp/eigprof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 /tmp/eigprof.log > gpurun_out/eig_timeline.md
python tools/eig_timeline.py $(find gpurun_out/prof_eig -name "*.db" | head -1) gpurun_out/eig_timeline_body.md > /dev/null 2>&1
cat gpurun_out/eig_timeline_body.md >> gpurun_out/eig_timeline.md
rm -rf gpurun_out/prof_eig gpurun_out/eig_timeline_body.md
cat gpurun_out/eig_timeline.md
