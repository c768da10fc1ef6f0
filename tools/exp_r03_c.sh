mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/gram_fuse_bench.py 2>&1 | tail -4 | tee gpurun_out/exp_r03_c.log
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "gram" 2>&1 | tail -3 | tee -a gpurun_out/exp_r03_c.log
