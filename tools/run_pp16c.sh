cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pixel_pair" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_sweep_nhwc.py -x -q 2>&1 | tail -2
for c in 1 0 1 0; do LK_PIXPAIR16=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline_families']; print('PP16', $c, round(d['value']), round(d['ms_per_step'],3), {k:round(f[k]['ms_per_step'],3) for k in f if k.startswith('pixpair')})"; done
