"""One warm-up fit, then ONE K-minibatch fit (accumulator creation ... finalize) — the driver's timed region — for a
kernel trace of what a short fit pays besides its minibatches.  usage: fit_tail.py [K]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda")) for _ in range(4)]
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = b.kron_accumulator(50000)
    for i in range(K):
        acc.add_batch(*data[i % 4])
    if os.environ.get("LK_SYNC_BEFORE_FINALIZE"):
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss, H = acc.finalize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"fit {rep}: {1e3 * (t2 - t0):.1f} ms = {1e3 * (t2 - t0) / K:.2f} per step (host enqueue of the minibatches {1e3 * (t1 - t0):.1f} ms)", flush=True)
    del H
