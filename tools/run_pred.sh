cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/kron_predictive_c4.py 2>&1 | tail -1 | cut -c1-300; done
