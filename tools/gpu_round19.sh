#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log gpurun_out/pmc
for inner in 3 2 1; do
  LK_EIG_INNER=$inner timeout 300 python tools/eig_study.py > gpurun_out/eig_study_$inner.log 2>&1
  echo "eig study inner=$inner rc=$?" >> gpurun_out/summary.log
done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
done
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py gpurun_out/pmc/pmc_FETCH_SIZE_results.db gpurun_out/pmc/pmc_WRITE_SIZE_results.db gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic.md > /dev/null 2>> gpurun_out/summary.log
ls gpurun_out/pmc >> gpurun_out/summary.log
rm -rf gpurun_out/pmc
tail -n 8 gpurun_out/eig_study_3.log | cut -c1-600; cat gpurun_out/summary.log
