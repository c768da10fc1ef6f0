mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_n.log; : > $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_sweep_nhwc.py -m gpu -q -x > gpurun_out/exp_r03_n_tests.log 2>&1
echo "tests rc=$?" >> $O; tail -3 gpurun_out/exp_r03_n_tests.log | cut -c1-250 >> $O
B=$GRAFT_REPO_ROOT/laplace_amd/csrc/liblaplace_hip_b.so
for rep in 1 2 3; do
  echo "A (default lib): $(timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O
  echo "B (variant lib): $(LK_LIB=$B timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O
done
cat $O
