set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_sweep_nhwc.py -x -q 2>&1 | tail -5
timeout 300 python tools/conv_tile_sweep.py 2>&1 | tail -22
for c in 2 24578 2 24578; do LK_CONV_CONFIG=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CFG', $c, d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"; done
