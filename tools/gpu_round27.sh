#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/summary.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gram or shift" > gpurun_out/t_gram.log 2>&1
echo "gram tests rc=$?" >> gpurun_out/summary.log
for bk in 16 32; do
LK_GRAM_BK=$bk timeout 300 python tools/microbench.py gram 2>&1 | grep -v "Cannot find" > gpurun_out/mb_gram_pipe_bk$bk.log
echo "microbench bk=$bk rc=${PIPESTATUS[0]}" >> gpurun_out/summary.log
done
tail -1 gpurun_out/t_gram.log
cat gpurun_out/summary.log
