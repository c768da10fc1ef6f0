cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_backend.py -x -q 2>&1 | tail -3
for c in 1 0 1 0; do LK_LAG_JOIN=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LAG', $c, round(d['value']), round(d['ms_per_step'],3), d.get('dropin_fit_samples_per_s'))"; done
