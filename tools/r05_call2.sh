#!/bin/bash
# round 5, GPU call 2: window-staging / column co-location variants of the persistent kernel (stand-alone and in the step),
# the per-image forward after the bn_act change, small configs with stacked minibatches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/winp_variants.py --json gpurun_out/r05_winp_variants.json > gpurun_out/r05_winp_variants.log 2>&1
tail -60 gpurun_out/r05_winp_variants.log
timeout 200 python -m pytest tests/test_gpu_per_image.py tests/test_gpu_conv.py -m gpu -q -x > gpurun_out/r05_call2_tests.log 2>&1
tail -3 gpurun_out/r05_call2_tests.log
for cfg in 2 1073741826 268435458 536870914 2; do
  LK_CONV_CONFIG=$cfg timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-predictive --no-eigh --no-extras --no-check > gpurun_out/r05_call2_bench_$cfg.json 2> gpurun_out/r05_call2_bench_$cfg.err
  python - <<P
import json
try:
    d = json.loads(open("gpurun_out/r05_call2_bench_$cfg.json").read().strip().splitlines()[-1])
    f = d["roofline_families"]
    print("config $cfg", round(d["value"]), round(d["ms_per_step"], 3), {k: round(f[k]["ms_per_step"], 3) for k in ("convp16", "bnact16", "conv16") if k in f})
except Exception as e:
    print("bench parse", e)
P
done
timeout 300 python tools/small_configs.py > gpurun_out/r05_small_configs.log 2>&1; tail -12 gpurun_out/r05_small_configs.log
