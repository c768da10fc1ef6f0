"""Exact diagonal GGN fit on ResNet-18 (config c4's model, batch 128) on the MI355X — development tool: whole-model
`backend.diag` rate with the Jacobian-free kernel (`lk_diag_ggn_shared_f32`), and for two layer shapes the kernel
against the route it replaced (`lk_jac_conv_f32` materialising [B, S, Do*Dk] + `lk_sq_colsum_f32`)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN  # noqa: E402
from laplace_amd._lib import get_kernels  # noqa: E402
from laplace_amd.nets import ResNet18  # noqa: E402

dev = "cuda"
torch.manual_seed(711)
model = ResNet18(10).to(dev).eval()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        for p in m.parameters():
            p.requires_grad_(False)
backend = HipGGN(model, "classification")
g = torch.Generator().manual_seed(0)
X = torch.randn(128, 3, 32, 32, generator=g).to(dev)
y = torch.randint(10, (128,), generator=g).to(dev)
out = {}
for _ in range(2):
    backend.diag(X, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    loss, h = backend.diag(X, y)
torch.cuda.synchronize()
out["diag_fit_samples_per_s"] = 5 * 128 / (time.perf_counter() - t0)
out["finite"] = bool(torch.isfinite(h).all())

K = get_kernels()
B, S = 128, 9
for name, Cin, Do, HW in (("layer1 3x3 64->64 @32x32", 64, 64, 32), ("layer4 3x3 512->512 @4x4", 512, 512, 4)):
    a = torch.randn(B, Cin, HW, HW, device=dev)
    gg = torch.randn(S, B, Do, HW, HW, device=dev)
    Dk, L = Cin * 9, HW * HW
    u = gg.reshape(S, B, Do, L).permute(1, 0, 2, 3).contiguous()
    v = torch.nn.functional.unfold(a, 3, 1, 1, 1).contiguous()
    h1 = torch.zeros(Do * Dk, device=dev)
    h0 = torch.zeros(Do * Dk, device=dev)

    def new():
        K.diag_ggn_shared(u, v, 1.0, h1)

    def old():
        Jl = torch.zeros(B, S, Do * Dk, device=dev)
        K.jac_conv(a, gg, (3, 3), (1, 1), (1, 1), (1, 1), Jl, 0, -1)
        K.sq_colsum(Jl, 0, Do * Dk, 1.0, h0)

    res = {}
    for fn, key in ((new, "shared_ms"), (old, "jacobian_route_ms")):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        res[key] = (time.perf_counter() - t0) / 3 * 1e3
    res["rel_diff"] = float(((h1 / 4) - (h0 / 4)).abs().max() / (h0 / 4).abs().max())
    res["TFLOPs_shared"] = 2.0 * B * S * Do * Dk * L / res["shared_ms"] / 1e9
    out[name] = res
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag_c4.json", "w"), indent=1)
