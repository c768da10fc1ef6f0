"""Throughput of the reference's literal fit loop `H += backend.kron(X, y, N)` (what `Laplace(..., backend=HipGGN).fit`
executes) next to the fused accumulator on the c4 workload (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN  # noqa: E402
from laplace_amd.kron import HipKron  # noqa: E402
from laplace_amd.nets import ResNet18  # noqa: E402

dev = "cuda"
torch.manual_seed(711)
model = ResNet18().to(dev).eval()
b = HipGGN(model, "classification")
X, y = torch.randn(128, 3, 32, 32, device=dev), torch.randint(0, 10, (128,), device=dev)
params = [p for p in model.parameters() if p.requires_grad]


def literal(steps):
    H = HipKron.init_from_model(params, dev, torch.float32)
    loss = 0
    for _ in range(steps):
        lb, Hb = b.kron(X, y, N=50000)
        loss += lb
        H += Hb
    return H


def fused(steps):
    acc = b.kron_accumulator(50000)
    for _ in range(steps):
        acc.add_batch(X, y)
    return acc.finalize()[1]


for name, fn in (("literal", literal), ("fused", fused)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / 20 * 1e3:.2f} ms/step, {128 * 20 / dt:.0f} samples/s", flush=True)
