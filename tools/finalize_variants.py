"""The end of a 20-minibatch fit (the driver's timed region) under the accumulator's finalize-time switches: early A-side
flush on / off, number of flush streams, a host sync in front of finalize.  ms per step over the whole fit, best of 5."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.backend import KronAccumulator
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda")) for _ in range(8)]
b = HipGGN(model, "classification")
K = 20
for name, kw, sync in (("default", {}, False), ("sync before finalize", {}, True), ("early_flush off", {"early_flush": False}, False),
                       ("flush_streams 1", {"flush_streams": 1}, False), ("flush_streams 6", {"flush_streams": 6}, False),
                       ("early off + sync", {"early_flush": False}, True), ("default", {}, False)):
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = b.kron_accumulator(50000)
        for k, v in kw.items():
            setattr(acc, k, v)
        for i in range(K):
            acc.add_batch(*data[i % 8])
        if sync:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        acc.finalize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t2 - t0, t2 - t1))
    st = torch.cuda.memory_stats()
    print("   device allocs so far %d, frees %d, reserved %.0f GiB" % (st["num_device_alloc"], st["num_device_free"], st["reserved_bytes.all.current"] / 2 ** 30))
    ts = ts[1:]
    print("%-22s: fit %.2f ms/step (min %.2f), finalize call to end %.1f ms" % (name, 1e3 * sum(t[0] for t in ts) / len(ts) / K, 1e3 * min(t[0] for t in ts) / K, 1e3 * sum(t[1] for t in ts) / len(ts)), flush=True)
