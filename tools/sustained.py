"""Steady-state c4 fit steps for a long stretch, timed in chunks: does the rate hold under sustained load?  (development tool)"""
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
chunks, per = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 200
for _ in range(4): acc.add_batch(X, y)
torch.cuda.synchronize()
t_start = time.perf_counter()
for c in range(chunks):
    t0 = time.perf_counter()
    for _ in range(per): acc.add_batch(X, y)
    th = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"t={t1 - t_start:6.1f}s  chunk {c:2d}: {1e3 * (t1 - t0) / per:.2f} ms/step (host enqueue {1e3 * (th - t0) / per:.2f})", flush=True)
