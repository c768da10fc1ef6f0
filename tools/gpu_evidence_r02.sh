# Round-2 secondary measurements with the final kernels (writes gpurun_out/; copied into profiles/r02_* afterwards)
mkdir -p gpurun_out
timeout 200 python tools/quadconv_bench.py > gpurun_out/qc.log 2>&1; echo "qc rc=$?" > gpurun_out/summary_ev2.log
timeout 200 python tools/small_configs.py > gpurun_out/small.log 2>&1; echo "small rc=$?" >> gpurun_out/summary_ev2.log
timeout 300 python tools/c5_bert.py > gpurun_out/c5.log 2>&1; echo "c5 rc=$?" >> gpurun_out/summary_ev2.log
timeout 200 python tools/diag_c4.py > gpurun_out/diag_c4.log 2>&1; echo "diag rc=$?" >> gpurun_out/summary_ev2.log
timeout 200 python tools/kron_predictive_c4.py > gpurun_out/kronpred.log 2>&1; echo "kronpred rc=$?" >> gpurun_out/summary_ev2.log
timeout 200 python tools/conv_f16x2_bench.py > gpurun_out/conv_bench.log 2>&1; echo "conv rc=$?" >> gpurun_out/summary_ev2.log
timeout 100 python tools/gram16_bench.py > gpurun_out/gram16_bench.log 2>&1
cat gpurun_out/summary_ev2.log; tail -1 gpurun_out/qc.log | cut -c1-300; tail -1 gpurun_out/small.log | cut -c1-600; tail -1 gpurun_out/c5.log | cut -c1-600; tail -1 gpurun_out/diag_c4.log | cut -c1-300; tail -1 gpurun_out/kronpred.log
