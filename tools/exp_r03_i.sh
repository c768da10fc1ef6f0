mkdir -p gpurun_out; export TMPDIR=/tmp
LK_LIB=$GRAFT_REPO_ROOT/laplace_amd/csrc/liblaplace_hip_dev.so timeout 600 python tools/win_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_r03_i.log
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "window" 2>&1 | tail -3
