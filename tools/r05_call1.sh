#!/bin/bash
# round 5, GPU call 1: the per-image forward kernels (new tests), the tests of everything they touch, a bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_per_image.py tests/test_gpu_dynamic_range.py tests/test_gpu_conv.py \
  tests/test_gpu_sweep_nhwc.py tests/test_gpu_timed_config.py tests/test_gpu_baseline_parity.py tests/test_gpu_fullsize.py \
  -m gpu -q -x -s --durations=8 > gpurun_out/r05_call1_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05_call1_tests.log
tail -25 gpurun_out/r05_call1_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_call1_bench.json 2> gpurun_out/r05_call1_bench.err
echo "bench rc $?"
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r05_call1_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step") if k in d}, d.get("roofline", {}).get("frac"))
except Exception as e:
    print("bench parse", e)
P
tail -5 gpurun_out/r05_call1_bench.err
