mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -q 2>&1 | tail -30 | cut -c1-250 | tee gpurun_out/exp_r03_p.log
