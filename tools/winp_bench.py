"""Persistent window kernel against the generic form (config bit 27 = persistent form off): fused backward-data launches of the c4
layer shapes, correctness against the generic kernel's result and time per launch.  Development tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = "cuda"
WP = 1 << 27


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = [(64, 64, 32, 1152)] + ([(128, 128, 16, 1152), (256, 256, 8, 1152), (512, 512, 4, 1152)] if "--all" in sys.argv else [])
for Co, Ci, H, N in cases:  # conv Ci -> Co; backward-data: cotangent [N, H, H, Co] -> [N, H, H, Ci]
    torch.manual_seed(0)
    m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).to(dev)   # backward-data: GEMM K = Co, GEMM N = Ci
    g = K.split_f16x2((torch.randn(N, H, H, Co, device=dev) * 1e-3).contiguous())
    add = K.split_f16x2((torch.randn(N, H, H, Ci, device=dev) * 1e-2).contiguous())
    S = 9 if N % 9 == 0 else 1
    mask = (torch.rand(N // S, H, H, Ci, device=dev) > 0.5).to(torch.uint8)
    prep = cv.PreparedConv(m)
    gf = 2.0 * N * H * H * Co * Ci * 9 / 1e9
    K.conv_config = 2 | WP
    ref = cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
    variants = [("generic", 2 | WP), ("persistent", 2), ("split", 2 | (1 << 25)), ("wide512", 2 | (1 << 26))]
    variants += [("stagger%d" % k, 2 | ((k + 1) << 20)) for k in (3,)]
    variants += [("coloc%d" % (1 << k), 2 | (k << 28)) for k in (1, 2, 3)] + [("persistent", 2)]  # (bits 28-29: log2 of the columns per XCD; 0 = default)
    for name, cfg in variants:  # (bit 27 switches the persistent form OFF)
        K.conv_config = cfg
        out = cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
        torch.cuda.synchronize()
        d = (out.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        am = abs(out.amax.item() - out.float().abs().max().item()) / out.amax.item()
        t = timeit(lambda: cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask))
        print(f"Ci={Ci} Co={Co} {H}x{H} N={N} {gf:.0f} GFLOP | {name:10s} {t:7.1f} us  {gf / t * 1e3:6.1f} TFLOP/s   rel diff vs generic {d:.2e}  sexp {int(out.sexp)} vs {int(ref.sexp)}  amax err {am:.1e}", flush=True)
    K.conv_config = 2
