cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "fused or position" 2>&1 | tail -2
for c in 2 2097154 2 2097154; do LK_CONV_CONFIG=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CFG', $c, round(d['value']), round(d['ms_per_step'],3), round(d['roofline_families']['conv16']['ms_per_step'],3))"; done
