"""cProfile of the host side of steady-state c4 fit steps (development tool: the step is within 15 - 20 % of host-bound)."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
acc = b.kron_accumulator(50000)
for _ in range(8): acc.add_batch(X, y)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n): acc.add_batch(X, y)
pr.disable()
th = time.perf_counter()
torch.cuda.synchronize()
print(f"{n} steps under cProfile: host {1e3 * (th - t0) / n:.2f} ms/step, wall {1e3 * (time.perf_counter() - t0) / n:.2f}")
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
