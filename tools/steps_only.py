"""N steady-state c4 fit steps and nothing else (for kernel traces): ResNet-18, batch 128, fused accumulator.
usage: steps_only.py [N] [serial]   (serial: one stream, no overlap of the factor kernels, one lane)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
serial = "serial" in sys.argv[2:]
acc = b.kron_accumulator(50000, overlap=not serial)
if serial:
    acc.lanes = 1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for _ in range(4): acc.add_batch(X, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): acc.add_batch(X, y)
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step")
