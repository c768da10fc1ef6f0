"""Ablation timing of the split-fp16 convolution kernels (development): which part of a launch costs what."""
import json, os, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels
K = get_kernels()
N = 1152
SHAPES = [(64, 64, 3, 1, 1, 32), (128, 128, 3, 1, 1, 16), (512, 512, 3, 1, 1, 4)]
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for cin, cout, k, s, p, H in SHAPES:
    m = nn.Conv2d(cin, cout, k, s, p, bias=False).cuda()
    g = torch.randn(N, H, H, cout, device="cuda")
    gs = K.split_f16x2(g)
    prep = cv.PreparedConv(m)
    out = torch.empty(N, H, H, cin, device="cuda")
    row = {"shape": [cin, cout, H]}
    for form, base in (("patch", 0), ("generic", 2)):
        for name, ab in (("full", 0), ("nostore", 1), ("nomfma", 2), ("nostage", 4), ("nomfma_nostore", 3), ("nostage_nostore", 5)):
            K.conv_config = base | (ab << 8)
            row[f"{form}_{name}"] = round(timeit(lambda: cv.conv_backward_data(prep, gs, (H, H), out=out)), 4)
    print(json.dumps(row))
