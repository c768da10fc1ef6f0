cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backend.py -x -q 2>&1 | tail -2
for c in 1 0; do LK_LAZY_KRON=$c timeout 200 python tools/literal_loop_bench.py 2>&1 | tail -2; done
