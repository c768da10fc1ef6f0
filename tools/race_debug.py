"""Is the reverse pass deterministic, and does the side-stream overlap perturb it?  (development tool)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.capture import Tape
from laplace_amd.nets import ResNet18

dev = "cuda"
torch.manual_seed(711)
model = ResNet18(10).to(dev).eval()
b = HipGGN(model, "classification")
g = torch.Generator().manual_seed(1)
X = torch.randn(16, 3, 32, 32, generator=g).to(dev)
y = torch.randint(10, (16,), generator=g).to(dev)
N = 50000

def grads():
    tape = Tape(model, b.params)
    f = tape.forward(X)
    p = torch.softmax(f.detach(), -1)
    S = torch.diag_embed(p.sqrt()) - p.unsqueeze(2) * p.sqrt().unsqueeze(1)
    out = tape.output_grads(f, S.permute(2, 0, 1).contiguous())
    names = [t.name for t in tape.taps]
    acts = [t.a.clone() for t in tape.taps]
    tape.release()
    return names, out, acts

def cmp(tag, A, B, names):
    worst = max(((a - c).abs().max() / c.abs().max()).item() for a, c in zip(A, B))
    bad = [(n, f"{((a - c).abs().max() / c.abs().max()).item():.1e}") for n, a, c in zip(names, A, B) if ((a - c).abs().max() / c.abs().max()).item() > 1e-6]
    print(tag, f"worst {worst:.1e}", bad[:6], flush=True)

names, g1, a1 = grads()
_, g2, a2 = grads()
cmp("A: grads run1 vs run2 (no side stream yet)", g1, g2, names)
cmp("A: acts  run1 vs run2", a1, a2, names)
for overlap in (False, True):
    acc = b.kron_accumulator(N, overlap=overlap)
    acc.add_batch(X, y)
    torch.cuda.synchronize()
    _, g3, a3 = grads()
    cmp(f"B: grads after add_batch(overlap={overlap}) vs run1", g3, g1, names)
    cmp(f"B: acts  after add_batch(overlap={overlap}) vs run1", a3, a1, names)
    lf, Hf = acc.finalize()
    ll, Hl = b.kron(X, y, N=N)
    worst = max(((u - v).abs().max() / v.abs().max()).item() for F1, F2 in zip(Hf.kfacs, Hl.kfacs) for u, v in zip(F1, F2))
    print(f"C: fused(overlap={overlap}) vs literal worst {worst:.1e}", flush=True)
# repeat literal twice
l1, H1 = b.kron(X, y, N=N)
l2, H2 = b.kron(X, y, N=N)
worst = max(((u - v).abs().max() / v.abs().max()).item() for F1, F2 in zip(H1.kfacs, H2.kfacs) for u, v in zip(F1, F2))
print(f"D: literal vs literal worst {worst:.1e}", flush=True)
