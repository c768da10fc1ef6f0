# rocprofv3 kernel trace of 12 fit steps (tools/host_bound.py's loop) -> gpurun_out/timeline_<tag>.md
TAG=${1:-x}
export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/steps_only.py 16 > $GRAFT_REPO_ROOT/gpurun_out/tl_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/tl_$TAG -name "*.db" | head -1)
python tools/timeline.py $DB gpurun_out/timeline_$TAG.md
cp $DB gpurun_out/timeline_$TAG.db; rm -rf gpurun_out/tl_$TAG
