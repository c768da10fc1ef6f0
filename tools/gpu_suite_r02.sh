# Round-2 evidence pass on the MI355X box: default bench, rocprofv3 kernel trace of the bench command, PMC traffic passes
# (FETCH_SIZE / WRITE_SIZE separately).  Writes under gpurun_out/; copy what is to be judged into profiles/.
TAG=${1:-v2}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" > gpurun_out/summary_$TAG.log
bash tools/prof_bench.sh $TAG > /dev/null 2>&1; echo "trace rc=$?" >> gpurun_out/summary_$TAG.log
rm -rf gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary_$TAG.log
  cd $GRAFT_REPO_ROOT
done
python tools/pmc_traffic.py $(find gpurun_out/pmc -name "*FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmc -name "*WRITE_SIZE*.db" | head -1) gpurun_out/pmc_traffic_$TAG.json gpurun_out/pmc_traffic_$TAG.md > /dev/null 2>> gpurun_out/summary_$TAG.log
python - <<'PY' >> gpurun_out/summary_$TAG.log 2>&1
import sqlite3, glob
db = glob.glob("gpurun_out/pmc/**/*FETCH_SIZE*.db", recursive=True)[0]
con = sqlite3.connect(db)
print([r[1] for r in con.execute("pragma table_info(pmc_events)")])
for r in list(con.execute("select * from pmc_events where name like '%conv_f16x2%' limit 12")): print(r)
PY
rm -rf gpurun_out/pmc
cat gpurun_out/summary_$TAG.log | cut -c1-300; tail -c 2500 gpurun_out/bench_$TAG.log
