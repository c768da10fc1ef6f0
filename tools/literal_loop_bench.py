"""The reference's literal fit loop on c4 (ResNet-18, batch 128): ``H += backend.kron(X, y, N)[1]`` per minibatch
(laplace/baselaplace.py:969-985), read into the public layout at the end — with the pixel-pair products left to the
running sum (default) and with every minibatch computing its own A factors (`backend.lazy_pixpair = False`), beside the fused
accumulator, all in one process.  usage: literal_loop_bench.py [n_minibatches]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN, HipKron
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
data = [(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda")) for _ in range(8)]
params = [p for p in model.parameters() if p.requires_grad]


def literal(k):
    H = HipKron.init_from_model(params, "cuda", torch.float32)
    for i in range(k):
        H += b.kron(*data[i % 8], N=50000)[1]
    return H.kfacs


def fused(k):
    acc = b.kron_accumulator(50000)
    for i in range(k):
        acc.add_batch(*data[i % 8])
    return acc.finalize()[1].kfacs


def timed(f):
    f(9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = f(n)
    torch.cuda.synchronize()
    return 128 * n / (time.perf_counter() - t0), out


ref_rate, ref = timed(fused)
print(f"fused accumulator           {ref_rate:9.0f} samples/s ({n} minibatches, finalize included)")
for flag in (True, False, True, False):
    b.lazy_pixpair = flag
    rate, got = timed(literal)
    err = max(float((x - y).abs().max() / (y.abs().max() + 1e-30)) for F, G in zip(got, ref) for x, y in zip(F, G))
    print(f"literal loop, deferred={int(flag)}   {rate:9.0f} samples/s   max rel diff to fused {err:.1e}")
