"""A few launches of the 64-channel fused backward-data layer through the generic and the window kernels (for PMC passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels
K = get_kernels()
dev = "cuda"
WIN, Z, W4 = 1 << 22, 1 << 23, 1 << 24
for C, H in ((64, 32), (128, 16)):
    torch.manual_seed(0)
    m = nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
    N = 1152
    g = K.split_f16x2((torch.randn(N, H, H, C, device=dev) * 1e-3).contiguous())
    add = K.split_f16x2((torch.randn(N, H, H, C, device=dev) * 1e-2).contiguous())
    mask = (torch.rand(128, H, H, C, device=dev) > 0.5).to(torch.uint8)
    prep = cv.PreparedConv(m)
    for kcfg in (2, 2 | WIN | W4, 2 | WIN | (Z if C == 64 else 0)):
        K.conv_config = kcfg
        for _ in range(4):
            cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
        torch.cuda.synchronize()
