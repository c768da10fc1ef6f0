mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_t.log; : > $O
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "window" 2>&1 | tail -2 >> $O
python tools/gram_fuse_bench.py > /dev/null 2>&1
for rep in 1 2; do
for cfg in "LK_CONV_CONFIG=2" "LK_CONV_CONFIG=12582914" "LK_CONV_CONFIG=20971522"; do
  echo "$cfg: $(env $cfg timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O
done; done
cat $O
