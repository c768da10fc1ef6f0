mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-a}
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_$TAG.json 2>gpurun_out/r04_bench_$TAG.err
echo "rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r04_bench_$TAG.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['step_breakdown'])
print('check', d['check']['max_block_rel_err'], d['check']['ok'], 'roofline', round(d['roofline']['frac'],3), d['roofline']['ms_per_step'])
print({k:(round(v['ms_per_step'],3),round(v['frac'],3)) for k,v in d['roofline_families'].items()})
print(d.get('fit_50k')); print(d.get('fit_fixed_cost')); print('eigh', d.get('eigh_ms'), 'dropin', d.get('dropin_fit_samples_per_s'))
print('pred', {k:v for k,v in d.get('predictive_kron_c4',{}).items() if k in ('samples_per_s','ms_per_call')})
PY
