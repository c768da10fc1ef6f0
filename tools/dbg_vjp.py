import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laplace_amd._lib import get_kernels
K = get_kernels()
torch.manual_seed(0)
S, B, H, W, C = 3, 5, 6, 6, 64
DEV = "cuda"
g = torch.randn(S * B, H, W, C, device=DEV) * 3.0
g2f = torch.randn(S * B, H, W, C, device=DEV) * 0.01
g2 = K.split_f16x2(g2f)
mask = torch.rand(B, H, W, C, device=DEV) > 0.4
multf = torch.randn(B, H, W, C, device=DEV) * 2.0
scale = (torch.rand(C, device=DEV) + 0.5) * torch.where(torch.rand(C, device=DEV) > 0.5, 1.0, -1.0)
s_amax = K.absmax(scale)
shape = (S * B, H, W, C)
print("scale amax word", s_amax.item(), scale.abs().max().item(), "g2 sexp", g2.sexp.item(), "g2 max", g2f.abs().max().item())
for name, sc in (("signed", scale), ("abs", scale.abs().contiguous())):
    out = K.vjp_nhwc_split(None, None, g2, None, None, sc, K.absmax(sc), S, shape)
    ref = g2.float().double() * sc.double()
    o = out.float().double()
    d = (o - ref).abs()
    i = d.argmax()
    print(name, "sexp", out.sexp.item(), "rel", (d.max() / ref.abs().max()).item(), "at", i.item(), "ref", ref.flatten()[i].item(),
          "got", o.flatten()[i].item(), "h", out.planes[0].flatten()[i].item(), "l", out.planes[1].flatten()[i].item(),
          "scale there", sc[i % C].item(), "g2 there", g2.float().flatten()[i].item())
    bad = (d > 1e-6 * ref.abs().max()).sum().item()
    print("   elements off by >1e-6 rel:", bad, "of", d.numel())
