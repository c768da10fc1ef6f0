cd $GRAFT_REPO_ROOT
CFGS=2,1048578 timeout 300 python tools/conv_f16x2_bench.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if 'shape' in d: print(d['shape'], round(d['ours2_ms']*1e3), round(d['ours1048578_ms']*1e3))
        else: print(d)
"
for c in 2 1048578 2 1048578; do LK_CONV_CONFIG=$c timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predictive --no-eigh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CFG', $c, round(d['value']), round(d['ms_per_step'],3), round(d['roofline_families']['conv16']['ms_per_step'],3))"; done
