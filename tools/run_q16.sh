cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "quadform_shared" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
for c in 1 0 1; do LK_QUAD16=$c timeout 300 python tools/kron_predictive_c4.py 2>&1 | tail -1 | cut -c1-400; done
