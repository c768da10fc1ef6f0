mkdir -p gpurun_out
python tools/count_calls.py 2>&1 | grep -v amdgpu | tee gpurun_out/exp_r03_o.log
