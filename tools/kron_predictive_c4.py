"""GLM predictive variance under the full-network KFAC posterior of config c4 (ResNet-18) on the MI355X (development
tool): the weight-sharing quadratic-form kernel against the as-written route (assemble each layer's Jacobian block,
rotate it with two GEMMs, matrix.py:406-461) that it replaces.  `--profile` runs only the fused route (for rocprofv3)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN  # noqa: E402
from laplace_amd import predictive as P  # noqa: E402
from laplace_amd._lib import get_kernels  # noqa: E402
from laplace_amd.nets import ResNet18  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--profile", action="store_true")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--calls", type=int, default=-1, help="set up, one warm-up call, then exactly this many fused calls and exit (PMC passes: tools/gpu_evidence.sh)")
args = ap.parse_args()
dev = "cuda"
torch.manual_seed(711)
model = ResNet18(10).to(dev).eval()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        for p in m.parameters():
            p.requires_grad_(False)
backend = HipGGN(model, "classification")
g = torch.Generator().manual_seed(0)
X = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
y = torch.randint(10, (args.batch,), generator=g).to(dev)
acc = backend.kron_accumulator(50000)
for _ in range(2):
    acc.add_batch(X, y)
_, H = acc.finalize()
post = H.decompose() + torch.ones(1, device=dev)
K = get_kernels()
out = {"batch": args.batch}
if args.calls >= 0:
    P.glm_variance_kron(backend, X, post)
    for _ in range(args.calls):
        P.glm_variance_kron(backend, X, post)
    torch.cuda.synchronize()
    sys.exit(0)


def rate(x, reps):
    P.glm_variance_kron(backend, x, post)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f_mu, f_var = P.glm_variance_kron(backend, x, post)
    torch.cuda.synchronize()
    return reps * len(x) / (time.perf_counter() - t0), f_var


out["fused_samples_per_s"], fv = rate(X, 5)
K.profile = prof0 = {}
P.glm_variance_kron(backend, X, post)
torch.cuda.synchronize()
K.profile = None
out["families_ms"] = {k: round(sum(e[0].elapsed_time(e[1]) for e in v), 3) for k, v in prof0.items()}
if hasattr(K, "use_quad_planes"):  # the fp32-operand route (rotations emit fp32, in-flight three-piece bf16 split) beside it
    K.use_quad_planes = False
    out["fp32_operand_route_samples_per_s"], fv32 = rate(X, 5)
    K.profile = prof1 = {}
    P.glm_variance_kron(backend, X, post)
    torch.cuda.synchronize()
    K.profile = None
    out["families_ms_fp32_operand_route"] = {k: round(sum(e[0].elapsed_time(e[1]) for e in v), 3) for k, v in prof1.items()}
    out["planes_vs_fp32_route_max_rel_diff"] = float((fv - fv32).abs().max() / fv32.abs().max())
    K.use_quad_planes = True
if not args.profile:
    small = X[:8]
    _, fv_small = rate(small, 1)
    K.quadform_shared_max_outputs = 0  # as written: Jacobian block + two rotations per layer
    try:
        out["as_written_samples_per_s_batch8"], fv_ref = rate(small, 1)
        out["max_rel_diff_vs_as_written"] = float((fv_small - fv_ref).abs().max() / fv_ref.abs().max())
    finally:
        K.quadform_shared_max_outputs = 10
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
if not args.profile:
    json.dump(out, open("gpurun_out/kron_predictive_c4.json", "w"), indent=1)
