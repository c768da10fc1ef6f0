# eigensolver concurrency: hardware-queue count (GPU_MAX_HW_QUEUES) x stream count, and a kernel trace of decompose
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/eig_queues.log; : > $O
for q in default 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> $O
  if [ $q = default ]; then timeout 200 python tools/eig_streams.py 2>&1 | grep n_streams >> $O
  else GPU_MAX_HW_QUEUES=$q timeout 200 python tools/eig_streams.py 2>&1 | grep n_streams >> $O; fi
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/eigprof -o eig -- python $GRAFT_REPO_ROOT/tools/eig_streams.py > /tmp/eigprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/eigprof -name '*kernel_stats.csv' | head -1)
echo "== kernel stats ($f)" >> $O
python - "$f" >> $O <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'eig' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>8s} total ms {float(r['TotalDurationNs'])/1e6:9.1f} avg us {float(r['AverageNs'])/1e3:8.1f}")
P
cat $O
