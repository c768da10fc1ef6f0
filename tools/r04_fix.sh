# round 4, fixed-cost work: new kernels + the timed configuration on the device, then the tail trace and the driver's bench
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_timed_config.py tests/test_lazy_kron.py tests/test_gpu_switches.py -m gpu -x -q -k "two_accumulator or finalize_factors or pixel_pair or timed or lazy or LANES or FLUSH or lanes" > gpurun_out/r04_fix_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r04_fix_tests.log
bash tools/r04_tail.sh b > gpurun_out/r04_fix_tail.log 2>&1
grep -v "^W2026\|^E2026" gpurun_out/fit_tail_b.log | tail -4
head -3 gpurun_out/fit_tail_timeline_b.md
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_fix_bench.json 2>gpurun_out/r04_fix_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_fix_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['step_breakdown'], d['check'], d.get('fit_50k'))
PY
