mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "two_accumulator or finalize_factors or pixel_pair" 2>&1 | tail -3
timeout 300 python tools/count_sites.py > gpurun_out/r04_count_sites.txt 2>&1; tail -45 gpurun_out/r04_count_sites.txt
timeout 300 python tools/finalize_cost.py 20 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-check > gpurun_out/r04_fix2_bench.json 2>gpurun_out/r04_fix2_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_fix2_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('fit_50k'))
PY
