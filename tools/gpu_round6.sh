#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.log
for t in test_c4_resnet18_kfac_additivity_and_traces test_c4_fused_accumulator_equals_literal_loop test_c4_eigendecomposition_round_trip test_c2_lenet_fused_predictive_equals_materialised test_c3_last_layer_dense_predictive; do
  timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -k $t > gpurun_out/t_$t.log 2>&1
  echo "$t rc=$?" >> gpurun_out/summary.log
done
cat gpurun_out/summary.log
