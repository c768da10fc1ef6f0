mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_q.log; : > $O
for rep in 1 2; do
for cfg in "LK_CONV_CONFIG=2" "LK_CONV_CONFIG=33554434" "LK_CONV_CONFIG=524290"; do
  echo "$cfg: $(env $cfg timeout 300 python tools/steps_only.py 48 2>&1 | tail -1)" >> $O
done; done
timeout 600 python -m pytest tests/test_gpu_switches.py -m gpu -q -k "chunk" 2>&1 | tail -2 >> $O
cat $O
