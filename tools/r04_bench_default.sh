mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-v2}
timeout 900 python bench.py > gpurun_out/r04_bench_$TAG.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/r04_bench_$TAG.log | tail -1 > gpurun_out/r04_bench_c4_$TAG.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_c4_$TAG.json").read())
print(round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), d["fit_50k"], d["fit_fixed_cost"])
PY
