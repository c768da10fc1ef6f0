# Round-3 evidence pass on the MI355X box: the whole -m gpu suite, the default bench line, the rocprofv3 kernel trace of
# the bench command and of the steady-state step, and the PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately, stamped
# with the hash of the kernel sources).  Writes under gpurun_out/; copy what is to be judged into profiles/.
TAG=${1:-v1}
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/r03_summary_$TAG.log; : > $S
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=10 > gpurun_out/r03_gpu_tests_$TAG.log 2>&1; echo "tests rc=$?" >> $S
tail -4 gpurun_out/r03_gpu_tests_$TAG.log >> $S
timeout 900 python bench.py > gpurun_out/r03_bench_$TAG.log 2>&1; echo "bench rc=$?" >> $S
grep '^{' gpurun_out/r03_bench_$TAG.log | tail -1 > gpurun_out/r03_bench_c4_$TAG.json
ARGS="--steps 10 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check"
rm -rf gpurun_out/prof_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo "trace rc=$?" >> $GRAFT_REPO_ROOT/$S
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/r03_bench_c4_kernel_stats_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_$TAG
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/steps_only.py 16 > $GRAFT_REPO_ROOT/gpurun_out/prof_steps_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) gpurun_out/r03_steps_only_kernel_stats_$TAG.md > /dev/null 2>&1
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $GRAFT_REPO_ROOT/$S
  cd $GRAFT_REPO_ROOT
done
python tools/pmc_traffic.py $(find gpurun_out/pmc -name "*FETCH_SIZE*.db" | head -1) $(find gpurun_out/pmc -name "*WRITE_SIZE*.db" | head -1) gpurun_out/r03_pmc_traffic_bench_c4_$TAG.json gpurun_out/r03_pmc_traffic_bench_c4_$TAG.md > /dev/null 2>> $S
rm -rf gpurun_out/pmc
cat $S; tail -c 1500 gpurun_out/r03_bench_$TAG.log
