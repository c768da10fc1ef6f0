"""Throughput of the small BASELINE configs on the MI355X (development tool): c1 MLP 1-50-1 regression
(diag + kron), c2 LeNet-5 KFAC-GGN.  Whole `fit` incl. decomposition, samples/s."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd.laplace import HipLaplace  # noqa: E402
from laplace_amd.nets import lenet5, mlp_1_50_1  # noqa: E402

dev = "cuda"
out = {}


def run(name, model, X, y, lik, hs, bs, reps=3):
    la = None
    ts = []
    loader = [(X[i:i + bs], y[i:i + bs]) for i in range(0, len(X), bs)]

    class L(list):
        dataset = X

    loader = L(loader)
    for r in range(reps + 1):
        la = HipLaplace(model, lik, "all", hs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        la.fit(loader)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[len(ts[1:]) // 2]
    out[name] = {"fit_ms": t * 1e3, "samples_per_s": len(X) / t, "first_ms": ts[0] * 1e3, "batches": len(loader)}
    print(name, out[name], flush=True)
    return la


torch.manual_seed(711)
m = mlp_1_50_1().to(dev)
X = (8 * torch.rand(1000, 1)).to(dev)
y = (torch.sin(X) + 0.3 * torch.randn_like(X)).to(dev)
run("c1_diag", m, X, y, "regression", "diag", 100)
run("c1_kron", m, X, y, "regression", "kron", 100)
run("c1_full", m, X, y, "regression", "full", 100)
torch.manual_seed(711)
m = lenet5().to(dev)
X = torch.randn(10000, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (10000,), device=dev)
la = run("c2_kron", m, X, y, "classification", "kron", 256)
run("c2_diag", m, X, y, "classification", "diag", 256)
# predictive on c2 (steady state: the first calls pay MIOpen's solver search for the new shapes)
for _ in range(2):
    la._glm_predictive_distribution(X[:256])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(0, 2048, 256):
    la._glm_predictive_distribution(X[i:i + 256])
torch.cuda.synchronize()
out["c2_kron_predictive_samples_per_s"] = 2048 / (time.perf_counter() - t0)
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/small_configs.json", "w"), indent=1)
