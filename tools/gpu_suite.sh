#!/bin/bash
# Full GPU pass on the MI355X box (run through tools/gpurun_retry.sh): parity tests, default bench, rocprofv3 kernel
# trace + the two PMC traffic passes over the bench command, smoke().  Writes under gpurun_out/; copy what should be
# judged into profiles/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log gpurun_out/prof gpurun_out/pmc
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof/r6_results.db gpurun_out/stats_r6.md > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc/pmc_FETCH_SIZE_results.db gpurun_out/pmc/pmc_WRITE_SIZE_results.db gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic.md > /dev/null 2>> gpurun_out/summary.log
rm -rf gpurun_out/prof/*.db gpurun_out/pmc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/summary.log
tail -3 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log | cut -c1-300; tail -1 gpurun_out/smoke.log; cat gpurun_out/summary.log
