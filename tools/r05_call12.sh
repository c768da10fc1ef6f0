#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_quad_planes.py -m gpu -q > gpurun_out/r05_call12_tests.log 2>&1; tail -4 gpurun_out/r05_call12_tests.log
timeout 300 python tools/kron_predictive_c4.py --profile > gpurun_out/r05_pred_planes.log 2>&1; tail -2 gpurun_out/r05_pred_planes.log
