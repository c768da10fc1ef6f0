"""Wall time of Kron.decompose on real c4 factors for different stream counts of the batched eigensolver (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
acc = b.kron_accumulator(50000)
for i in range(6):
    X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
    acc.add_batch(X, y)
_, H = acc.finalize()
torch.cuda.synchronize()
for ns in (6, 3, 4, 8, 12, 6):
    t0 = time.perf_counter()
    D = H.decompose(n_streams=ns)
    D.check_converged()
    torch.cuda.synchronize()
    print(f"n_streams {ns}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
