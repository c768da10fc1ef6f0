# rocprofv3 kernel trace of tools/fit_tail.py (a K-minibatch fit incl. finalize) -> gpurun_out/fit_tail_<tag>.md
TAG=${1:-x}; K=${2:-20}
export TMPDIR=/tmp
rm -rf /tmp/ft_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/ft_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/fit_tail.py $K > $GRAFT_REPO_ROOT/gpurun_out/fit_tail_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/fit_tail_timeline.py $(find /tmp/ft_$TAG -name "*.db" | head -1) gpurun_out/fit_tail_$TAG.md > /dev/null 2>&1
rm -rf /tmp/ft_$TAG
cat gpurun_out/fit_tail_$TAG.log | grep fit
