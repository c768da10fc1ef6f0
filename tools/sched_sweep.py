"""Steady-state c4 step time under the accumulator's scheduling attributes (lanes, lag of the factor kernels, queue priority of
the lanes, pixel-pair group size) — re-measured in round 6 because the step's bound changed (power) since they were tuned."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96


def run(attrs):
    b = HipGGN(model, "classification")
    best = 1e9
    for rep in range(3):
        acc = b.kron_accumulator(50000)
        for k, v in attrs.items():
            setattr(acc, k, v)
        for _ in range(16):
            acc.add_batch(X, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            acc.add_batch(X, y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
        acc.finalize()
    return best


base = run({})
print("default: %.2f ms per step" % base, flush=True)
for attrs in ({"lanes": 1}, {"lanes": 3}, {"lag_join": False}, {"lag_depth": 2}, {"lag_depth": 3}, {"lane_priority": 0}, {"pix_group": 4},
              {"pix_group": 16}, {"max_ahead": 1}, {"max_ahead": 4}, {"lanes": 3, "lag_depth": 2}, {}):
    print("%-34s %.2f ms per step" % (str(attrs), run(attrs)), flush=True)
