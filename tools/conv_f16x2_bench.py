"""Per-layer timing of the split-fp16 backward-data convolution (lk_conv_nhwc_f16x2) against MIOpen's fp32 backward-data
on the shapes of the c4 sweep (ResNet-18, batch 9 seeds x 128 = 1152).  Development tool; writes
gpurun_out/conv_f16x2_bench.json."""
import json
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import conv as cv  # noqa: E402
from laplace_amd._lib import get_kernels  # noqa: E402

SHAPES = [  # (Cin, Cout, k, stride, pad, Hin, count in ResNet-18)
    (64, 64, 3, 1, 1, 32, 4), (64, 128, 3, 2, 1, 32, 1), (128, 128, 3, 1, 1, 16, 3), (64, 128, 1, 2, 0, 32, 1),
    (128, 256, 3, 2, 1, 16, 1), (256, 256, 3, 1, 1, 8, 3), (128, 256, 1, 2, 0, 16, 1), (256, 512, 3, 2, 1, 8, 1),
    (512, 512, 3, 1, 1, 4, 3), (256, 512, 1, 2, 0, 8, 1),
]
N = int(os.environ.get("NB", 1152))
dev = "cuda"
K = get_kernels()
rows = []
CFGS = [int(c) for c in os.environ.get("CFGS", "2,0").split(",")]
tot = {**{f"ours{c}": 0.0 for c in CFGS}, "miopen": 0.0}


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for cin, cout, k, s, p, H, cnt in SHAPES:
    m = nn.Conv2d(cin, cout, k, s, p, bias=False).to(dev)
    Ho = (H + 2 * p - k) // s + 1
    g = torch.randn(N, cout, Ho, Ho, device=dev)
    gs = K.split_f16x2(g.permute(0, 2, 3, 1).contiguous())
    prep = cv.PreparedConv(m)
    out = torch.empty(N, H, H, cin, device=dev)
    flop = 2.0 * N * Ho * Ho * cout * cin * k * k
    res = {"shape": [cin, cout, k, s, H], "gflop": flop / 1e9}
    for cfg in CFGS:  # 2: generic form (default); 0: patch form where eligible
        K.conv_config = cfg
        ms = timeit(lambda: cv.conv_backward_data(prep, gs, (H, H), out=out))
        res[f"ours{cfg}_ms"] = ms
        res[f"ours{cfg}_tf"] = flop / ms / 1e9
        tot[f"ours{cfg}"] += cnt * ms
    dummy = g.new_empty(N, cin, H, H)
    ms = timeit(lambda: torch.ops.aten.convolution_backward(g, dummy, m.weight, None, m.stride, m.padding, m.dilation, False,
                                                           [0, 0], 1, [True, False, False]))
    res["miopen_ms"], res["miopen_tf"] = ms, flop / ms / 1e9
    tot["miopen"] += cnt * ms
    # split cost (what a producer would fold into its own pass)
    xf = g.permute(0, 2, 3, 1).contiguous()
    res["split_ms"] = timeit(lambda: K.split_f16x2(xf))
    rows.append(res)
    print(json.dumps(res))
print(json.dumps({"per_step_ms": tot, "batch": N}))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"rows": rows, "per_step_ms": tot, "batch": N}, open("gpurun_out/conv_f16x2_bench.json", "w"), indent=1)
