export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 2 1; do
echo "== workgroups per CU: $w"
LK_WINP_WGS=$w timeout 200 python tools/winp_bench.py 2>&1 | grep "persistent"
rm -rf $R/gpurun_out/pmcw
(cd /tmp && LK_WINP_WGS=$w timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcw -o c -- python $R/tools/winp_pmc_run.py > $R/gpurun_out/pmcw_f.log 2>&1)
python $R/tools/rocpd_pmc.py $R/gpurun_out/r04_pmc_winp_f$w.md $(find $R/gpurun_out/pmcw -name "*.db") > /dev/null 2>&1
rm -rf $R/gpurun_out/pmcw
grep "conv_winp" $R/gpurun_out/r04_pmc_winp_f$w.md | cut -c1-200
done
