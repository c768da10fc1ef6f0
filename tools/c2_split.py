"""c2 (LeNet-5 KFAC, N = 10 000, batch 256): where the whole-fit time goes — accumulate / finalize / decompose."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.laplace import HipLaplace
from laplace_amd.nets import lenet5

torch.manual_seed(711)
m = lenet5().cuda().eval()
X = torch.randn(10000, 3, 32, 32, device="cuda")
y = torch.randint(0, 10, (10000,), device="cuda")
b = HipGGN(m, "classification")


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    t0 = sync()
    acc = b.kron_accumulator(10000)
    for i in range(0, 10000, 256):
        acc.add_batch(X[i:i + 256], y[i:i + 256])
    t1 = sync()
    _, H = acc.finalize()
    t2 = sync()
    D = H.decompose()
    t3 = sync()
    D.check_converged()
    t4 = sync()
    print(f"accumulate {1e3 * (t1 - t0):.1f} ms  finalize {1e3 * (t2 - t1):.1f}  decompose {1e3 * (t3 - t2):.1f}  check {1e3 * (t4 - t3):.1f}  "
          f"sizes {sorted({M.shape[0] for F in H.kfacs for M in F})}", flush=True)


class L(list):
    dataset = X


loader = L([(X[i:i + 256], y[i:i + 256]) for i in range(0, 10000, 256)])
for rep in range(3):
    la = HipLaplace(m, "classification", "all", "kron")
    t0 = sync()
    la.fit(loader)
    t1 = sync()
    print(f"HipLaplace.fit {1e3 * (t1 - t0):.1f} ms", flush=True)
