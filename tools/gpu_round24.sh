#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log gpurun_out/stress_*.log
for name in a b; do
  timeout 300 python -X faulthandler tools/stress_abort.py 40 2>&1 | grep -v "Cannot find the function" > gpurun_out/stress_$name.log
  echo "stress $name rc=${PIPESTATUS[0]}" >> gpurun_out/summary.log
done
LK_SWEEP=0 timeout 300 python -X faulthandler tools/stress_abort.py 40 2>&1 | grep -v "Cannot find the function" > gpurun_out/stress_nosweep.log
echo "stress nosweep rc=${PIPESTATUS[0]}" >> gpurun_out/summary.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/summary.log
tail -3 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log | cut -c1-300; tail -2 gpurun_out/smoke.log; cat gpurun_out/summary.log
