mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/exp_r03_e.log; : > $O
for cfg in "LK_FUSE_GRAM=0 LK_CONV_CONFIG=2" "LK_FUSE_GRAM=0 LK_CONV_CONFIG=0" "LK_FUSE_GRAM=0 LK_CONV_CONFIG=2 LK_NO_OVERLAP=1" "LK_FUSE_GRAM=0 LK_CONV_CONFIG=0 LK_NO_OVERLAP=1" "LK_FUSE_GRAM=1 LK_CONV_CONFIG=2 LK_NO_OVERLAP=1" "LK_FUSE_GRAM=0 LK_CONV_CONFIG=524290" "LK_FUSE_GRAM=0 LK_CONV_CONFIG=2 LK_PIX_GROUP=16"; do
  for rep in 1 2; do echo "$cfg: $(env $cfg python tools/steps_only.py 48 2>&1 | tail -1)" >> $O; done
done
cat $O
