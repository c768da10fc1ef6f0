# rocprofv3 kernel trace of tools/winp_one.py per (shape, config): average duration of the persistent window kernel itself
# usage: bash tools/winp_prof.sh <tag> "<Ci Co H N>" cfg1 cfg2 ...   -> gpurun_out/winp_prof_<tag>.log
TAG=$1; SHAPE=$2; shift 2
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/winp_prof_$TAG.log
for CFG in "$@"; do
  D=/tmp/wp_$TAG_$CFG; rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -o p -- python $GRAFT_REPO_ROOT/tools/winp_one.py $SHAPE $CFG 20 > /dev/null 2>&1)
  DB=$(find $D -name "*.db" | head -1)
  echo "shape $SHAPE config $CFG: $(cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py $DB 2>/dev/null | grep winp | cut -d'|' -f3-7)" >> $OUT
  rm -rf $D
done
