# PMC traffic passes + kernel traces + the bench line on the FINAL kernel sources (the table's stamp must match them).
TAG=${1:-v3}
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
S=gpurun_out/r03_summary_$TAG.log; : > $S
ARGS="--steps 10 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check"
rm -rf gpurun_out/pmc gpurun_out/prof_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $GRAFT_REPO_ROOT/$S
  cd $GRAFT_REPO_ROOT
done
python tools/pmc_traffic.py $(find gpurun_out/pmc -name "*FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmc -name "*WRITE_SIZE*.db" | head -1) gpurun_out/r03_pmc_traffic_bench_c4_$TAG.json gpurun_out/r03_pmc_traffic_bench_c4_$TAG.md > /dev/null 2>> $S
rm -rf gpurun_out/pmc
cp gpurun_out/r03_pmc_traffic_bench_c4_$TAG.json profiles/   # (so that the bench run below finds a table with a matching stamp)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo "trace rc=$?" >> $GRAFT_REPO_ROOT/$S
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/r03_bench_c4_kernel_stats_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_$TAG
timeout 900 python bench.py > gpurun_out/r03_bench_$TAG.log 2>&1; echo "bench rc=$?" >> $S
grep '^{' gpurun_out/r03_bench_$TAG.log | tail -1 > gpurun_out/r03_bench_c4_$TAG.json
python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
cat $S; python - <<PY
import json
d=json.loads(open("gpurun_out/r03_bench_c4_$TAG.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:120])
PY
