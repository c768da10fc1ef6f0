"""Marginal-likelihood prior sweep on the MI355X (development tool): 100 Adam steps on the log prior precision
(scalar and per-layer) over a posterior with config c4's block structure (ResNet-18: 20 conv + fc weight blocks and
the fc bias block), the whole-posterior logdet (`lk_kron_logdet_blocks_f32`) against the block-by-block one."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd.kron import HipKronDecomposed  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
G = [64] * 5 + [128] * 5 + [256] * 5 + [512] * 5 + [10]
A = [27] + [576] * 4 + [576, 1152, 1152, 1152, 64] + [1152, 2304, 2304, 2304, 128] + [2304, 4608, 4608, 4608, 256] + [512]
vals = [[(torch.rand(a, generator=g) + 1e-3).to(dev), (torch.rand(b, generator=g) * 5).to(dev)] for a, b in zip(G, A)]
vals.append([(torch.rand(10, generator=g) + 1e-3).to(dev)])
vecs = [[torch.empty(0, device=dev) for _ in ls] for ls in vals]  # logdet never touches the eigenvectors
H = HipKronDecomposed(vecs, vals)
n_terms = sum(len(ls[0]) * (len(ls[1]) if len(ls) == 2 else 1) for ls in vals)
out = {"blocks": len(vals), "log_terms": n_terms}


def sweep(per_layer, steps=100):
    log_pp = torch.zeros(len(vals) if per_layer else 1, device=dev, requires_grad=True)
    opt = torch.optim.Adam([log_pp], lr=0.1)
    hf = torch.tensor(1.0, device=dev)
    for it in range(steps + 10):
        if it == 10:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.zero_grad()
        pp = log_pp.exp()
        neg = 0.5 * ((H * hf + pp).logdet() - (pp.log().sum() if per_layer else pp.log() * n_terms).sum())
        neg.backward()
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, float(neg.detach())


for fused in (True, False):
    HipKronDecomposed.fused_logdet = fused
    for per_layer in (False, True):
        ms, val = sweep(per_layer)
        out[f"{'fused' if fused else 'blockwise'}_{'layerwise' if per_layer else 'scalar'}_100_steps_ms"] = ms
        out[f"{'fused' if fused else 'blockwise'}_{'layerwise' if per_layer else 'scalar'}_final"] = val
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/marglik_bench.json", "w"), indent=1)
