"""Per-layer timing of the weight-sharing predictive kernel (`lk_kron_quadform_shared_f32`) on the layer shapes of
config c4 (ResNet-18, 10 outputs, batch 128) — development tool."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels  # noqa: E402

K = get_kernels()
dev = "cuda"
B, C = 128, 10
shapes = [("conv1", 64, 27, 1024), ("layer1", 64, 576, 1024), ("layer2.0", 128, 576, 256), ("layer2", 128, 1152, 256),
          ("layer2.sc", 128, 64, 256), ("layer3", 256, 2304, 64), ("layer3.sc", 256, 128, 64), ("layer4.0", 512, 2304, 16),
          ("layer4", 512, 4608, 16), ("layer4.sc", 512, 256, 16)]
out = {}
for name, Do, Dk, L in shapes:
    u = torch.randn(B, C, Do, L, device=dev)
    v = torch.randn(B, Dk, L, device=dev)
    l1, l2 = torch.rand(Do, device=dev), torch.rand(Dk, device=dev)
    d = torch.ones(1, device=dev)
    fv = torch.zeros(B, C, C, device=dev)
    for _ in range(2):
        K.kron_quadform_shared(u, v, l1, l2, d, fv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        K.kron_quadform_shared(u, v, l1, l2, d, fv)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * C * Do * Dk * L
    out[name] = {"Do": Do, "Dk": Dk, "L": L, "ms": ms, "TFLOPs": fl / ms / 1e9}
    print(name, out[name], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/quadconv_bench.json", "w"), indent=1)
