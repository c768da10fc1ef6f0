"""Does the driver's timed fit (20 minibatches behind a 5-minibatch warm-up fit) pay for allocator growth?  Per lead of the host
(`KronAccumulator.max_ahead`): a 5-minibatch fit, then three 20-minibatch fits, ms per step and reserved memory after each."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.backend import KronAccumulator
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda")) for _ in range(8)]
for ahead in [int(v) for v in (sys.argv[1:] or ["8", "4", "2", "8"])]:
    KronAccumulator.max_ahead = ahead
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    b = HipGGN(model, "classification")
    out = []
    for K in (5, 20, 20, 20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = b.kron_accumulator(50000)
        for i in range(K):
            acc.add_batch(*data[i % 8])
        acc.finalize()
        torch.cuda.synchronize()
        out.append("%d: %.2f ms/step (%.0f GiB)" % (K, 1e3 * (time.perf_counter() - t0) / K, torch.cuda.memory_reserved() / 2 ** 30))
    print("max_ahead %d | %s" % (ahead, " | ".join(out)), flush=True)
