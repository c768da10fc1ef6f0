#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/mask_flips.py 64 > gpurun_out/r05_mask_flips.log 2>&1; tail -25 gpurun_out/r05_mask_flips.log
timeout 900 python -m pytest tests/test_gpu_quad_planes.py tests/test_gpu_conv.py tests/test_gpu_per_image.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_parity.py -m gpu -q -x > gpurun_out/r05_call5_tests.log 2>&1; tail -6 gpurun_out/r05_call5_tests.log
timeout 300 python tools/kron_predictive_c4.py --profile > gpurun_out/r05_pred_planes.log 2>&1; tail -2 gpurun_out/r05_pred_planes.log
