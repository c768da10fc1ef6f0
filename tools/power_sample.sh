# rocm-smi power / clock samples while a load runs in the background (development tool): is the step bound by the chip's power?
# usage: bash tools/power_sample.sh <tag> <command...>   -> gpurun_out/power_<tag>.log
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/power_$TAG.log
: > $OUT
"$@" > $GRAFT_REPO_ROOT/gpurun_out/power_${TAG}_load.log 2>&1 &
PID=$!
sleep 25   # (import + warm-up)
for i in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|Temperature \(Sensor (junction|edge)" >> $OUT
  echo "--" >> $OUT
  sleep 1
done
wait $PID
echo "idle:" >> $OUT
/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $OUT
tail -3 $GRAFT_REPO_ROOT/gpurun_out/power_${TAG}_load.log >> $OUT
