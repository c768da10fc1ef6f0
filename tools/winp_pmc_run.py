"""A few launches of the fused 64-channel backward-data layer through the generic and the persistent window kernels (for PMC passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels
K = get_kernels(); dev = "cuda"
torch.manual_seed(0)
C, H, N = 64, 32, 1152
m = nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
g = K.split_f16x2((torch.randn(N, H, H, C, device=dev) * 1e-3).contiguous())
add = K.split_f16x2((torch.randn(N, H, H, C, device=dev) * 1e-2).contiguous())
mask = (torch.rand(128, H, H, C, device=dev) > 0.5).to(torch.uint8)
prep = cv.PreparedConv(m)
for kcfg in (2 | (1 << 27), 2):
    K.conv_config = kcfg
    for _ in range(4):
        cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
    torch.cuda.synchronize()
