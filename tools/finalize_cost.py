"""What a short fit pays besides its minibatches (the driver's bench is 20 steps): c4, K minibatches, then `finalize` —
accumulate / finalize wall time separately, for one and two lanes.  usage: finalize_cost.py [K]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda")) for _ in range(8)]


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for lanes in (2, 1, 2, 1):
    for rep in range(2):
        acc = b.kron_accumulator(50000)
        acc.lanes = lanes
        t0 = sync()
        for i in range(K):
            acc.add_batch(*data[i % 8])
        t1 = sync()
        acc.finalize()
        t2 = sync()
    print(f"lanes {lanes}: {K} minibatches {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / K:.2f} per step), finalize {1e3 * (t2 - t1):.1f} ms, "
          f"together {1e3 * (t2 - t0) / K:.2f} ms per step", flush=True)
