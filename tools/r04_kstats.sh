# kernel stats of 20 serial (no overlap) steady-state steps -> gpurun_out/r04_kstats_<tag>.md
TAG=${1:-x}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ks_$TAG
cd /tmp && LK_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/ks_$TAG -o p -- python $R/tools/steps_only.py 20 > $R/gpurun_out/ks_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/ks_$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r04_kstats_$TAG.md > /dev/null
rm -rf gpurun_out/ks_$TAG
tail -1 gpurun_out/ks_$TAG.log | head -1; grep wall gpurun_out/ks_$TAG.log
head -30 gpurun_out/r04_kstats_$TAG.md
