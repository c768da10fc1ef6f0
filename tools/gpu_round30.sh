#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
# occupancy experiment: BIG tile uses 33.8 KB; pad to force 2 / 1 workgroups per CU (160 KB LDS)
for pad in 0 20000 50000 100000; do
  LK_GRAM_LDS_PAD=$pad timeout 300 python tools/microbench.py gram 2>&1 | grep "'tn'\|conv_fused'" > gpurun_out/mb_pad$pad.log
  echo "pad $pad rc=${PIPESTATUS[0]}" >> gpurun_out/summary.log
done
cat gpurun_out/summary.log
