cd $GRAFT_REPO_ROOT
for g in 4 8 16 4 16; do LK_PIX_GROUP=$g timeout 300 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GROUP', $g, round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['roofline_families'].items()})"; done
