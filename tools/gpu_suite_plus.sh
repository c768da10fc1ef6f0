bash tools/gpu_suite.sh
timeout 600 python tools/kron_predictive_c4.py > gpurun_out/kronpred.log 2>&1
timeout 600 python tools/diag_c4.py > gpurun_out/diag_c4.log 2>&1
tail -1 gpurun_out/kronpred.log; tail -1 gpurun_out/diag_c4.log
