"""One fused backward-data shape on the persistent window kernel, `n` launches with one config word (development tool; meant to
run under `rocprofv3 --kernel-trace`: tools/winp_prof.sh).  argv: Ci Co H N config [n]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels

K = get_kernels()
Ci, Co, H, N, cfg = [int(v) for v in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
torch.manual_seed(0)
m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).cuda()
g = K.split_f16x2((torch.randn(N, H, H, Co, device="cuda") * 1e-3).contiguous())
add = K.split_f16x2((torch.randn(N, H, H, Ci, device="cuda") * 1e-2).contiguous())
mask = (torch.rand(N // 9 if N % 9 == 0 else N, H, H, Ci, device="cuda") > 0.5).to(torch.uint8)
prep = cv.PreparedConv(m)
K.conv_config = cfg
for _ in range(n):
    cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
torch.cuda.synchronize()
