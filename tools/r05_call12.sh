#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_quad_planes.py tests/test_gpu_conv.py -m gpu -q -x > gpurun_out/r05_call12_tests.log 2>&1; tail -6 gpurun_out/r05_call12_tests.log
timeout 300 python tools/kron_predictive_c4.py --profile > gpurun_out/r05_pred_planes.log 2>&1; tail -2 gpurun_out/r05_pred_planes.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_baseline_parity.py tests/test_gpu_switches.py -m gpu -q -x > gpurun_out/r05_call12_tests2.log 2>&1; tail -4 gpurun_out/r05_call12_tests2.log
