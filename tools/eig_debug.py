"""Per-factor eigensolver diagnostics on the actual ResNet-18 KFAC factors (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd._lib import get_kernels
from laplace_amd.nets import ResNet18

dev = "cuda"
torch.manual_seed(711)
model = ResNet18(10).to(dev).eval()
b = HipGGN(model, "classification")
acc = b.kron_accumulator(50000)
g = torch.Generator().manual_seed(1)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    X = torch.randn(128, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(10, (128,), generator=g).to(dev)
    acc.add_batch(X, y)
loss, H = acc.finalize()
K = get_kernels()
seen = set()
for bi, F in enumerate(H.kfacs):
    for fi, M in enumerate(F):
        n = M.shape[0]
        torch.cuda.synchronize()
        t0 = time.time()
        w, Q, info = K.syevj(M.contiguous(), max_sweeps=30)
        torch.cuda.synchronize()
        dt = (time.time() - t0) * 1e3
        M64 = M.double()
        M64 = torch.triu(M64) + torch.triu(M64, 1).T
        wref = torch.linalg.eigvalsh(M64).clamp(min=0)
        scale = wref.max().item()
        w64, Q64 = w.double(), Q.double()
        val = (w64 - wref).abs().max().item() / scale
        rec = ((Q64 * w64) @ Q64.T - M64).abs().max().item() / scale
        orth = (Q64.T @ Q64 - torch.eye(n, device=dev, dtype=torch.float64)).abs().max().item()
        rank = int((wref > 1e-7 * scale).sum().item())
        print(f"blk {bi}.{fi} n={n:5d} rank~{rank:5d} maxdiag/lmax={M.diagonal().max().item()/scale:.2e} "
              f"info={info.tolist()} ms={dt:8.1f} val={val:.1e} rec={rec:.1e} orth={orth:.1e}", flush=True)
