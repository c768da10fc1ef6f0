"""finalize() of a c4 fit behind a drained device: host time until the call returns against the time until the device is
done, and a cProfile of the call (development tool: is the once-per-fit finalisation bound by the host's launch rate?)."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(4):
    acc = b.kron_accumulator(50000)
    for _ in range(K):
        acc.add_batch(X, y)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if rep == 3 else None
    t0 = time.perf_counter()
    if pr: pr.enable()
    acc.finalize()
    if pr: pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"finalize: host {1e3 * (t1 - t0):.2f} ms, device done after {1e3 * (t2 - t0):.2f} ms")
    if pr:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
