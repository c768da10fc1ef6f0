# (like power_sample.sh, short warm-up) usage: bash tools/power_sample2.sh <tag> <warm seconds> <command...>
TAG=$1; WARM=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/power_$TAG.log
: > $OUT
"$@" > $GRAFT_REPO_ROOT/gpurun_out/power_${TAG}_load.log 2>&1 &
PID=$!
sleep $WARM
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 1
done
kill $PID 2>/dev/null; wait $PID 2>/dev/null
