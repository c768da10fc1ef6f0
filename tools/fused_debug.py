"""Which factor differs between the fused accumulator and the literal loop, and who is right (fp64)?"""
import os, sys
import torch
import torch.nn.functional as F
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
for overlap in (False, True):
    acc = b.kron_accumulator(N, overlap=overlap)
    acc.add_batch(X, y)
    lf, Hf = acc.finalize()
    ll, Hl = b.kron(X, y, N=N)
    # fp64 reference from the same activations / gradients
    tape = Tape(model, b.params)
    f = tape.forward(X)
    p = torch.softmax(f.detach().double(), -1)
    S = torch.diag_embed(p.sqrt()) - p.unsqueeze(2) * p.sqrt().unsqueeze(1)
    grads = tape.output_grads(f, S.permute(2, 0, 1).float().contiguous())
    blk = 0
    print("overlap", overlap)
    for tap, gg in zip(tape.taps, grads):
        Gf, Af = Hf.kfacs[blk]
        Gl, Al = Hl.kfacs[blk]
        blk += 2 if tap.has_bias else 1
        a = tap.a.double()
        if tap.kind == "conv2d":
            m = tap.module
            cols = F.unfold(a, m.kernel_size, dilation=m.dilation, padding=m.padding, stride=m.stride)
            L = cols.shape[-1]
            Aref = torch.einsum("bil,bjl->ij", cols, cols) / (N * L)
            g2 = gg.double().reshape(-1, gg.shape[2], L)
            Gref = torch.einsum("bil,bjl->ij", g2, g2)
        else:
            Aref = a.T @ a / N
            g2 = gg.double().reshape(-1, gg.shape[-1])
            Gref = g2.T @ g2
        r = lambda u, v: ((u.double() - v).abs().max() / v.abs().max()).item()
        print(f"{tap.name:28s} n_A={Af.shape[0]:5d} A fused-vs-ref {r(Af, Aref):.1e} lit-vs-ref {r(Al, Aref):.1e} | "
              f"G fused-vs-ref {r(Gf, Gref):.1e} lit-vs-ref {r(Gl, Gref):.1e}", flush=True)
    tape.release()
