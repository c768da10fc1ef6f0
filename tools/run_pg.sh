cd $GRAFT_REPO_ROOT
for g in 4 8 4 8; do LK_PIX_GROUP=$g timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline_families']; print('GROUP', $g, round(d['value']), round(d['ms_per_step'],3), {k:round(f[k]['ms_per_step'],3) for k in f if k.startswith('pixpair')})"; done
