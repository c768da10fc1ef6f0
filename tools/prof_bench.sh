# rocprofv3 kernel trace of the default bench command (few steps) -> gpurun_out/prof_<tag>.md  (usage: prof_bench.sh tag)
TAG=${1:-x}
export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/prof_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_$TAG
head -45 gpurun_out/prof_$TAG.md
