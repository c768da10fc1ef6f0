# Round-4 evidence pass on the MI355X box (FINAL kernel sources: the PMC table's stamp must match them):
#   1. the whole -m gpu suite; the parity numbers its rel() helpers measured -> gpurun_out/parity_errors.log
#   2. PMC traffic passes over the bench command (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only)
#   3. rocprofv3 --kernel-trace --stats of the driver's bench command
#   4. the bench lines: default flags and the driver's `--steps 20 --warmup 5`
#   5. smoke()
# Writes under gpurun_out/; copy what is to be judged into profiles/.   usage: bash tools/gpu_evidence_r04.sh [tag] [notests]
TAG=${1:-v1}
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=gpurun_out/r04_summary_$TAG.log; : > $S
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -rs --durations=10 > gpurun_out/r04_gpu_tests_$TAG.log 2>&1; echo "tests rc=$?" >> $S
  tail -4 gpurun_out/r04_gpu_tests_$TAG.log >> $S
  cp gpurun_out/parity_errors.log gpurun_out/r04_parity_errors_$TAG.log 2>/dev/null
fi
LIGHT="--no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check"
rm -rf gpurun_out/pmc gpurun_out/prof_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc -o pmc_$c -- python $R/bench.py --steps 2 --warmup 1 $LIGHT > $R/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $R/$S
  cd $R
done
python tools/pmc_traffic.py $(find gpurun_out/pmc -name "*FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmc -name "*WRITE_SIZE*.db" | head -1) gpurun_out/r04_pmc_traffic_bench_c4_$TAG.json gpurun_out/r04_pmc_traffic_bench_c4_$TAG.md > /dev/null 2>> $S
rm -rf gpurun_out/pmc
cp gpurun_out/r04_pmc_traffic_bench_c4_$TAG.json profiles/   # (so that the bench runs below find a table with a matching stamp)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py --steps 20 --warmup 5 $LIGHT > $R/gpurun_out/prof_$TAG.log 2>&1; echo "trace rc=$?" >> $R/$S
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/r04_bench_c4_kernel_stats_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_$TAG
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_k20_$TAG.log 2>&1; echo "bench k20 rc=$?" >> $S
grep '^{' gpurun_out/r04_bench_k20_$TAG.log | tail -1 > gpurun_out/r04_bench_c4_k20_$TAG.json
timeout 900 python bench.py > gpurun_out/r04_bench_$TAG.log 2>&1; echo "bench rc=$?" >> $S
grep '^{' gpurun_out/r04_bench_$TAG.log | tail -1 > gpurun_out/r04_bench_c4_$TAG.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $S 2>&1
cat $S; python - <<PY
import json
for f in ("gpurun_out/r04_bench_c4_k20_$TAG.json", "gpurun_out/r04_bench_c4_$TAG.json"):
    d=json.loads(open(f).read())
    print(f, round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), d["roofline"]["traffic"], str(d["roofline"].get("traffic_source"))[:100])
PY
