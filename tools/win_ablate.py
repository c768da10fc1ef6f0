"""Where the time of a fused backward-data launch goes (development build, LK_LIB=.../liblaplace_hip_dev.so): the 64- and
128-channel layers of c4 at batch 9 x 128, generic vs window kernels, with parts of the kernel switched off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = "cuda"
WIN, Z, W4 = 1 << 22, 1 << 23, 1 << 24
AB = {"full": 0, "no-epilogue": 256, "no-mfma": 512, "no-staging": 1024,
      "no-barrier": 1 << 21, "no-mfma+no-staging": 512 | 1024,
      "no-staging+no-barrier": 1024 | (1 << 21), "no-reads(+mfma)": 1 << 26, "no-reads+no-staging": 1024 | (1 << 26),
      "no-reads+no-staging+no-epilogue": 256 | 1024 | (1 << 26), "no-staging+no-epilogue": 256 | 1024,
      "no-mfma+no-staging+no-epilogue": 256 | 512 | 1024}


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


for C, H in ((64, 32), (128, 16)):
    torch.manual_seed(0)
    m = nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
    N = 1152
    g = K.split_f16x2((torch.randn(N, H, H, C, device=dev) * 1e-3).contiguous())
    add = K.split_f16x2((torch.randn(N, H, H, C, device=dev) * 1e-2).contiguous())
    mask = (torch.rand(128, H, H, C, device=dev) > 0.5).to(torch.uint8)
    prep = cv.PreparedConv(m)
    gf = 2.0 * N * H * H * C * C * 9 / 1e9
    for kname, kcfg in (("generic", 2), ("window 4 waves", 2 | WIN | W4), ("window 8 waves", 2 | WIN | (Z if C == 64 else 0))):
        row = []
        for aname, abit in AB.items():
            if kname == "generic" and (aname in ("no-barrier", "no-staging+no-barrier") or "reads" in aname):
                continue
            K.conv_config = kcfg | abit
            t = timeit(lambda: cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask))
            row.append(f"{aname} {t:.0f}")
        K.conv_config = 2
        print(f"C={C} {H}x{H} {gf:.0f} GFLOP | {kname:15s} | " + " | ".join(row) + " (us)", flush=True)
