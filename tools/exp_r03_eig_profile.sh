mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/eig_profile.log; : > $O
timeout 300 python tools/eig_profile.py 2>&1 | grep -v amdgpu.ids >> $O
rm -rf gpurun_out/prof_eig
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_eig -o p -- python $GRAFT_REPO_ROOT/tools/eig_profile.py decompose > /tmp/eigprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_eig -name "*.db" | head -1) gpurun_out/eig_kernel_stats.md > /dev/null 2>&1
grep -i "eig" gpurun_out/eig_kernel_stats.md | cut -c1-150 >> $O
rm -rf gpurun_out/prof_eig
cat $O
