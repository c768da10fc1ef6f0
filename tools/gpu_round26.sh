#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
for bk in 16 24 32; do
  LK_GRAM_BK=$bk timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gram or shift" > gpurun_out/t_gram_bk$bk.log 2>&1
  echo "gram tests bk=$bk rc=$?" >> gpurun_out/summary.log
  LK_GRAM_BK=$bk timeout 300 python tools/microbench.py gram 2>&1 | grep -v "Cannot find" > gpurun_out/mb_gram_bk$bk.log
  echo "microbench bk=$bk rc=${PIPESTATUS[0]}" >> gpurun_out/summary.log
done
for bk in 16 24 32; do tail -1 gpurun_out/t_gram_bk$bk.log; done
cat gpurun_out/summary.log
