TAG=${1:-a}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ft_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/ft_$TAG -o p -- python $R/tools/fit_tail.py ${2:-20} > $R/gpurun_out/fit_tail_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/ft_$TAG -name "*.db" | head -1)
python tools/fit_tail_timeline.py $DB gpurun_out/fit_tail_timeline_$TAG.md
tail -5 gpurun_out/fit_tail_$TAG.log
rm -rf gpurun_out/ft_$TAG
