"""Does the 64-byte-per-pixel granularity of a 32-channel K chunk cost bandwidth?  backward-data with the cotangent's
channel count (GEMM K per tap) = 32 (a pixel's chunk is the whole 64-byte row: contiguous) vs 64 / 128 (half / quarter
of the row per stage)."""
import json, os, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels
K = get_kernels()
N = 1152
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for cin, cout, H in ((64, 32, 32), (64, 64, 32), (64, 128, 32), (128, 32, 16), (128, 64, 16), (128, 128, 16), (128, 256, 16)):
    m = nn.Conv2d(cin, cout, 3, 1, 1, bias=False).cuda()
    g = torch.randn(N, H, H, cout, device="cuda")
    gs = K.split_f16x2(g)
    prep = cv.PreparedConv(m)
    out = torch.empty(N, H, H, cin, device="cuda")
    K.conv_config = 2
    ms = timeit(lambda: cv.conv_backward_data(prep, gs, (H, H), out=out))
    flop = 2.0 * N * H * H * cout * cin * 9
    print(json.dumps({"gemm_n": cin, "k_per_tap": cout, "H": H, "ms": round(ms, 4), "tf": round(flop / ms / 1e9, 1),
                      "us_per_stage_round": round(ms * 1e3 / (9 * cout / 32), 2)}))
