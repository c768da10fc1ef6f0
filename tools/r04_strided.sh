mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "strided" > gpurun_out/r04_strided_tests.log 2>&1
echo "rc=$?"; tail -15 gpurun_out/r04_strided_tests.log
timeout 900 python -m pytest tests/test_gpu_sweep_nhwc.py tests/test_gpu_timed_config.py tests/test_gpu_baseline_parity.py -m gpu -x -q > gpurun_out/r04_strided_tests2.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r04_strided_tests2.log
bash tools/r04_step_ab.sh LK_FUSE_STRIDED=0 LK_FUSE_STRIDED=1
