#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dict_inputs_c5.py tests/test_gpu_switches.py tests/test_gpu_per_image.py -m gpu -q -x > gpurun_out/r05_call3_tests.log 2>&1; tail -4 gpurun_out/r05_call3_tests.log
timeout 300 python tools/kron_predictive_c4.py --profile --quad16 > gpurun_out/r05_pred_ab.log 2>&1; tail -2 gpurun_out/r05_pred_ab.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_call3_bench.json 2> gpurun_out/r05_call3_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r05_call3_bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r05_call3_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"]), round(d["ms_per_step"], 3), {k: r.get(k) for k in ("family", "frac", "bound", "bound_from", "frac_hbm", "frac_of_fp32_mfma_peak", "algorithmic_bytes_per_launch")})
print(d["whole_step"]); print(d.get("predictive_samples_per_s")); print(d["predictive"]); print(d["other_configs"]); print(d["fit_fixed_cost"]); print(d.get("eigh_ms"), d["fit_50k"]["samples_per_s"])
P
