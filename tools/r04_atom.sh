export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q > gpurun_out/r04_atom_tests.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/r04_atom_tests.log
bash tools/r04_step_ab.sh LK_COPY_ABSMAX=0 LK_COPY_ABSMAX=1
