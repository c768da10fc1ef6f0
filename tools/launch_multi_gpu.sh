#!/bin/bash
# The driver's multi-GPU bench on ONE node, N = 2 4 8 (one process per GPU, RCCL over xGMI), with the environment the
# dmabuf-IPC-only host driver needs and the collective's self-checks switched on: bench.py asserts that the fit's single
# all-reduce moved exactly the bytes the factor shapes predict (ResNet-18: 188 MB of packed upper triangles) and reports
# its stand-alone time / bus bandwidth next to the fit rate.  usage: tools/launch_multi_gpu.sh [steps] [warmup]
set -u
STEPS=${1:-100}; WARM=${2:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-WARN}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
[ "$NGPU" -ge 2 ] || { echo "needs >= 2 GPUs (found $NGPU)"; exit 2; }
mkdir -p gpurun_out
python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-extras --no-cpu-baseline > gpurun_out/scale_1.json 2> gpurun_out/scale_1.err || exit 1
for N in 2 4 8; do
  [ "$N" -le "$NGPU" ] || break
  PORT=$((29500 + N))
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
    bench.py --gpus $N --steps $STEPS --warmup $WARM > gpurun_out/scale_$N.json 2> gpurun_out/scale_$N.err \
    || { echo "N=$N failed:"; tail -20 gpurun_out/scale_$N.err; exit 1; }
done
python - <<'PY'
import json, glob
rows = {}
for f in sorted(glob.glob("gpurun_out/scale_*.json")):
    line = [l for l in open(f) if l.startswith("{")]
    if line:
        d = json.loads(line[-1]); rows[d["n_gpus"]] = d
base = rows.get(1, {}).get("value")
for n, d in sorted(rows.items()):
    ar = d.get("allreduce") or {}
    print(f"N={n}: {d['value']:.0f} samples/s  x{d['value'] / base:.2f}" if base else f"N={n}: {d['value']:.0f}",
          f"all-reduce {ar.get('ms')} ms, {ar.get('bytes')} B (ok={ar.get('ok')}), eigh {d.get('eigh_ms')} ms")
PY
