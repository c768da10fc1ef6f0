export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_k20_v4.log 2>&1; echo "k20 rc=$?"
grep '^{' gpurun_out/r04_bench_k20_v4.log | tail -1 > gpurun_out/r04_bench_c4_k20_v4.json
timeout 600 python bench.py > gpurun_out/r04_bench_v4.log 2>&1; echo "default rc=$?"
grep '^{' gpurun_out/r04_bench_v4.log | tail -1 > gpurun_out/r04_bench_c4_v4.json
python - <<PY
import json
for f in ("gpurun_out/r04_bench_c4_k20_v4.json", "gpurun_out/r04_bench_c4_v4.json"):
    d=json.loads(open(f).read())
    print(f, round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), d["roofline"]["traffic"] is not None, d["fit_50k"]["accumulate_s"], d["fit_50k"]["samples_per_s"], d["fit_fixed_cost"]["finalize_ms_behind_a_drained_device"], d["step_breakdown"]["own_share"])
PY
