cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k fused_vjp 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_sweep_nhwc.py tests/test_gpu_baseline_parity.py -x -q 2>&1 | tail -5
for c in 1 0 1 0; do LK_FUSE_VJP=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FUSE', $c, round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['roofline_families'].items()})"; done
