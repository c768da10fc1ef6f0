export TMPDIR=/tmp
for rep in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-predictive --no-eigh --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K20', round(d['value']), round(d['ms_per_step'],3))"
done
