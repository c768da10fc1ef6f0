cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "quadform" 2>&1 | tail -2
for i in 1 2; do timeout 300 python tools/kron_predictive_c4.py 2>&1 | tail -1 | cut -c1-300; done
timeout 200 python tools/quadconv_bench.py 2>&1 | tail -3 | cut -c1-400
