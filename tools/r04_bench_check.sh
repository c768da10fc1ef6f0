export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_k20_v5.log 2>&1; echo "k20 rc=$?"
grep '^{' gpurun_out/r04_bench_k20_v5.log | tail -1 > gpurun_out/r04_bench_c4_k20_v5.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_c4_k20_v5.json").read())
print(round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), d["roofline"]["traffic"] is not None, d["fit_50k"]["samples_per_s"], d["predictive_kron_c4"]["cpu_baseline"], d["cpu_baseline"]["value"], sorted(d.keys()))
PY
tail -3 gpurun_out/r04_bench_k20_v5.log | cut -c1-300
