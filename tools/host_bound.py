"""Is the c4 step host-bound?  Enqueue time of a step (no synchronisation) vs its wall time on the GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
acc = b.kron_accumulator(50000)
for _ in range(4): acc.add_batch(X, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): acc.add_batch(X, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.2f} ms/step, wall {1e3*(t2-t0)/n:.2f} ms/step, drain after enqueue {1e3*(t2-t1):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): acc.add_batch(X, y)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
