import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels
K = get_kernels(); dev = "cuda"
WP = 1 << 27
Co, Ci, H, N = 64, 64, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1152
torch.manual_seed(0)
m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).to(dev)
g = K.split_f16x2((torch.randn(N, H, H, Co, device=dev) * 1e-3).contiguous())
prep = cv.PreparedConv(m)
K.conv_config = 2
add = K.split_f16x2((torch.randn(N, H, H, Ci, device=dev) * 1e-2).contiguous())
S = 9 if N % 9 == 0 else 1
mask = (torch.rand(N // S, H, H, Ci, device=dev) > 0.5).to(torch.uint8)
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
kw = {"add": add} if mode == "add" else {"mult": mask} if mode == "mask" else {"add": add, "mult": mask}
print("mode", mode)
ref = cv.conv_backward_data_vjp(prep, g, (H, H), **kw).float()
for rep in range(2):
    K.conv_config = 2
    out = cv.conv_backward_data_vjp(prep, g, (H, H), **kw).float()
    torch.cuda.synchronize()
    err = (out - ref).abs().reshape(-1, 64)          # [M, 64]
    bad = err.amax(1) > 1e-5 * ref.abs().max()
    idx = bad.nonzero().flatten()
    print(f"rep {rep}: bad rows {idx.numel()} of {bad.numel()}, max rel {float(err.max() / ref.abs().max()):.2e}")
    if idx.numel():
        tiles = torch.unique(idx // 256)
        print("  tiles:", tiles[:20].tolist(), "... n =", tiles.numel(), " tile % 512:", torch.unique(tiles % 512)[:20].tolist())
        print("  tile iteration (tile // 512):", torch.unique(tiles // 512).tolist())
        print("  row in tile:", torch.unique(idx % 256)[:40].tolist(), " n =", torch.unique(idx % 256).numel())
        print("  h:", torch.unique((idx % 1024) // 32).tolist(), " w:", torch.unique(idx % 32).tolist())
        r0 = int(idx[0]); print("  first bad row", r0, "channels bad:", (err[r0] > 1e-5 * ref.abs().max()).nonzero().flatten().tolist()[:16], float(out.reshape(-1,64)[r0,0]), float(ref.reshape(-1,64)[r0,0]))
K.conv_config = 2
