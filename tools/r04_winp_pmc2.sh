export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python tools/winp_bench.py 2>&1 | grep "persistent\|generic"
pass() {
  tag=$1; shift
  rm -rf $R/gpurun_out/pmcw
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmcw -o c -- python $R/tools/winp_pmc_run.py > $R/gpurun_out/pmcw_$tag.log 2>&1)
  python $R/tools/rocpd_pmc.py $R/gpurun_out/r04_pmc_winp_$tag.md $(find $R/gpurun_out/pmcw -name "*.db") > /dev/null 2>&1
  rm -rf $R/gpurun_out/pmcw
  grep "conv_\|kernel" $R/gpurun_out/r04_pmc_winp_$tag.md | cut -c1-300
}
pass d FETCH_SIZE
pass e WRITE_SIZE
