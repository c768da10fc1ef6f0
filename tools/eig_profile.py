"""Eigensolver on the real c4 KFAC factors (ResNet-18, 6 minibatches of 128): every distinct factor size solved alone
(ms, sweeps), then the whole `decompose` (43 factors).  `eig_profile.py decompose` runs only the latter (for kernel
traces)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd._lib import get_kernels
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
acc = b.kron_accumulator(50000)
for i in range(6):
    acc.add_batch(torch.randn(128, 3, 32, 32, device="cuda"), torch.randint(10, (128,), device="cuda"))
_, H = acc.finalize()
K = get_kernels()
torch.cuda.synchronize()
if len(sys.argv) < 2 or sys.argv[1] != "decompose":
    seen = {}
    for F in H.kfacs:
        for M in F:
            seen.setdefault(M.shape[0], M)
    for n in sorted(seen, reverse=True):
        M = seen[n].contiguous()
        K.syevj_batched([M])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        (w, Q, info), = K.syevj_batched([M])
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        rec = float(((Q * w) @ Q.T - M).abs().max() / M.abs().max())
        print(f"n {n:5d}: {ms:7.1f} ms  sweeps {int(info[1]):2d}  info {int(info[0])}  rec {rec:.1e}", flush=True)
for ns in (3, 2, 3, 2, 3):
    t0 = time.perf_counter()
    D = H.decompose(n_streams=ns)
    D.check_converged()
    torch.cuda.synchronize()
    print(f"decompose (n_streams {ns}): {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
