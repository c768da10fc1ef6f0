set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -5
CFGS=2,2050 timeout 300 python tools/conv_f16x2_bench.py 2>&1 | tail -12
for c in 2 2050 2; do LK_CONV_CONFIG=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CFG', $c, d['value'], d['ms_per_step'], d.get('step_breakdown'))"; done
