mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_l.log; : > $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_sweep_nhwc.py -m gpu -q -x > gpurun_out/exp_r03_l_tests.log 2>&1
echo "tests rc=$?" >> $O; tail -5 gpurun_out/exp_r03_l_tests.log | cut -c1-250 >> $O
for cfg in "LK_CONV_CONFIG=2" "LK_CONV_CONFIG=20971522" "LK_CONV_CONFIG=12582914"; do
  for rep in 1 2; do echo "$cfg: $(env $cfg timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O; done
done
cat $O
