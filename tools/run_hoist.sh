cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline_families']; print('RUN', round(d['value']), round(d['ms_per_step'],3), round(f['conv16']['ms_per_step'],3), round(f['conv16']['frac'],3))"; done
