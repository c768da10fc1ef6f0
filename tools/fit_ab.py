"""A/B of whole K-minibatch fits (the driver's timed region: accumulator creation ... finalize) in ONE process on one box,
alternating configurations.  usage: fit_ab.py [K] [reps] name=attr:value,... (attributes set on the accumulator)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
configs = []
for spec in sys.argv[3:] or ["default="]:
    name, _, rest = spec.partition("=")
    kv = {}
    for item in filter(None, rest.split(",")):
        a, _, v = item.partition(":")
        kv[a] = {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
    configs.append((name, kv))
data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda")) for _ in range(4)]


def fit(kv):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = b.kron_accumulator(50000)
    for a, v in kv.items():
        setattr(acc, a, v)
    for i in range(K):
        acc.add_batch(*data[i % 4])
    t1 = time.perf_counter()
    loss, H = acc.finalize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    fit.host = ((t1 - t0) * 1e3, (t2 - t1) * 1e3)
    return (time.perf_counter() - t0) * 1e3


for name, kv in configs:
    fit(kv)
res = {name: [] for name, _ in configs}
for r in range(reps):
    for name, kv in configs:
        res[name].append(fit(kv))
for name, v in res.items():
    print(f"(host: enqueue of the minibatches {fit.host[0]:.1f} ms, finalize call {fit.host[1]:.1f} ms)")
    print(f"{name:24s} K={K}: " + " ".join(f"{x:7.1f}" for x in v) + f"   best {min(v):7.1f} ms = {min(v) / K:.3f} ms/step", flush=True)
