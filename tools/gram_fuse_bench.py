"""Fused launch with / without the Gram of its result vs the stand-alone Gram kernel (64-channel 32 x 32 layer of c4,
batch 9 x 128).  Development tool."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = "cuda"
torch.manual_seed(0)
m = nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(dev)
S, B, H = 9, 128, 32
N = S * B
g = K.split_f16x2((torch.randn(N, H, H, 64, device=dev) * 1e-3).contiguous())
add = K.split_f16x2((torch.randn(N, H, H, 64, device=dev) * 1e-2).contiguous())
mask = (torch.rand(B, H, H, 64, device=dev) > 0.5).to(torch.uint8)
prep = cv.PreparedConv(m)
G = torch.zeros(64, 64, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def plain():
    return cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)


def fused():
    o = cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask, want_gram=True)
    K.gram_partials_reduce(o, 1.0, G)
    return o


def separate():
    o = cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
    K.gram_tn_f16x2(o, 1.0, G)
    return o


print(f"fused-epilogue conv alone        {timeit(plain):8.1f} us")
print(f"conv + Gram in the launch        {timeit(fused):8.1f} us")
print(f"conv, then lk_gram_tn_f16x2      {timeit(separate):8.1f} us")
