# fused Gram (C = 64) + range guard: tests, then the step with and without the fused Gram
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_b.log; : > $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_dynamic_range.py tests/test_gpu_timed_config.py tests/test_gpu_sweep_nhwc.py -m gpu -q -x -s -k "gram or dynamic or adversarial or range or per_image or saturated or timed or minibatch or consecutive or ragged or sweep" > gpurun_out/exp_r03_b_tests.log 2>&1
echo "tests rc=$?" >> $O
tail -30 gpurun_out/exp_r03_b_tests.log | cut -c1-300 >> $O
for cfg in "LK_FUSE_GRAM=1" "LK_FUSE_GRAM=0"; do
  for rep in 1 2; do echo "$cfg: $(env $cfg python tools/steps_only.py 48 2>&1 | tail -1)" >> $O; done
done
cat $O
