#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "syevj" > gpurun_out/t_eig.log 2>&1
echo "eig tests rc=$?" >> gpurun_out/summary.log
for ns in 6 12; do
  N_STREAMS=$ns timeout 300 python tools/eig_study.py > gpurun_out/eig_study_s$ns.log 2>&1
  echo "eig study streams=$ns rc=$?" >> gpurun_out/summary.log
done
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_all.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
tail -3 gpurun_out/t_eig.log; tail -n 3 gpurun_out/eig_study_s6.log | cut -c1-900; tail -n 1 gpurun_out/eig_study_s12.log | cut -c1-300; tail -3 gpurun_out/t_all.log; tail -1 gpurun_out/bench.log | cut -c1-300; cat gpurun_out/summary.log
