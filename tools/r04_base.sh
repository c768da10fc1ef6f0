# Round-4 baseline on today's box: the driver's bench command, steady-state steps, fixed cost split.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04_base.log; : > $O
echo "== steps_only 48" >> $O
timeout 300 python tools/steps_only.py 48 2>&1 | tail -1 >> $O
echo "== finalize_cost 20" >> $O
timeout 300 python tools/finalize_cost.py 20 2>&1 | tail -4 >> $O
echo "== bench --steps 20 --warmup 5" >> $O
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_base_bench.json 2>gpurun_out/r04_base_bench.err
python - <<'PY' >> $O
import json
d=json.loads(open('gpurun_out/r04_base_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['step_breakdown'], d['roofline']['frac'], d.get('eigh_ms'), d.get('fit_50k'))
print({k:(v['ms_per_step'],round(v['frac'],3)) for k,v in d['roofline_families'].items()})
PY
cat $O
