"""Per-launch times of the c4 Kron predictive's quadratic-form kernel (profile tag quadconv16) and rotation convolutions, in
launch order (development tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd import predictive as P
from laplace_amd._lib import get_kernels
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
backend = HipGGN(model, "classification")
g = torch.Generator().manual_seed(0)
X = torch.randn(128, 3, 32, 32, generator=g).cuda(); y = torch.randint(10, (128,), generator=g).cuda()
acc = backend.kron_accumulator(50000)
for _ in range(2):
    acc.add_batch(X, y)
_, H = acc.finalize()
post = H.decompose() + torch.ones(1, device="cuda")
K = get_kernels()
for _ in range(2):
    P.glm_variance_kron(backend, X, post)
torch.cuda.synchronize()
K.profile = prof = {}
P.glm_variance_kron(backend, X, post)
torch.cuda.synchronize()
K.profile = None
for key in ("quadconv16", "conv16"):
    evs = prof.get(key, [])
    print(key, "launches", len(evs), "total ms %.3f" % sum(e[0].elapsed_time(e[1]) for e in evs))
    for e in evs:
        ms = e[0].elapsed_time(e[1])
        print("   %.3f ms  %7.1f GFLOP  %6.1f TFLOP/s" % (ms, e[2] / 1e9, e[2] / ms / 1e9))
