#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "syevj" > gpurun_out/t_kernels.log 2>&1
echo "kernels(syevj) rc=$?" >> gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_backend.py tests/test_laplace_e2e.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_backend.log 2>&1
echo "backend+e2e rc=$?" >> gpurun_out/summary.log
timeout 900 python tools/microbench.py eig 64 128 256 576 1152 2304 4608 > gpurun_out/mb_eig.log 2>&1
echo "mb_eig rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/summary.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-predictive --no-eigh > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo "rocprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.log
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/t_kernels.log; tail -3 gpurun_out/t_backend.log; tail -2 gpurun_out/bench.log; cat gpurun_out/summary.log; ls -R gpurun_out/prof | head -20
