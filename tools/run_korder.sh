cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
CFGS=2,524290 timeout 300 python tools/conv_f16x2_bench.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if 'shape' in d: print(d['shape'], round(d['ours2_ms']*1e3), round(d['ours524290_ms']*1e3))
        else: print(d)
"
for c in 2 524290 2 524290; do LK_CONV_CONFIG=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CFG', $c, round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['roofline_families'].items()})"; done
