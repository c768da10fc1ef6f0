"""Tile-shape sweep of lk_conv_nhwc_f16x2 on the c4 layer shapes: backward-data at the sweep batch (9 seeds x 128) and
forward at batch 128, every explicit tile shape (config bits 12..14) against the default choice.  Development tool for
the tile heuristic; writes gpurun_out/conv_tile_sweep.json."""
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
TILES = {0: "auto", 6: "legacy", 1: "64x64", 2: "128x64", 3: "64x128", 4: "128x128", 5: "256x64"}
dev = "cuda"
K = get_kernels()


def timeit(fn, reps=10):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


rows = []
tot = {}
for direction, N in (("bwd", 1152), ("fwd", 128)):
    for cin, cout, k, s, p, H, cnt in SHAPES:
        m = nn.Conv2d(cin, cout, k, s, p, bias=False).to(dev)
        Ho = (H + 2 * p - k) // s + 1
        prep = cv.PreparedConv(m)
        flop = 2.0 * N * Ho * Ho * cout * cin * k * k
        if direction == "bwd":
            g = torch.randn(N, Ho, Ho, cout, device=dev)
            gs = K.split_f16x2(g)
            out = torch.empty(N, H, H, cin, device=dev)
            fn = lambda: cv.conv_backward_data(prep, gs, (H, H), out=out)
        else:
            x = torch.randn(N, H, H, cin, device=dev)
            xs = K.split_f16x2(x)
            out = torch.empty(N, Ho, Ho, cout, device=dev)
            fn = lambda: cv.conv_forward(prep, xs, out=out)
        res = {"dir": direction, "shape": [cin, cout, k, s, H], "gflop": round(flop / 1e9, 2)}
        for t, name in TILES.items():
            K.conv_config = 2 | (t << 12)
            ms = timeit(fn)
            res[name] = round(ms * 1e3, 1)  # us
            tot[(direction, name)] = tot.get((direction, name), 0.0) + cnt * ms
        best = min(TILES.values(), key=lambda n: res[n])
        res["best"] = best
        tot[(direction, "best")] = tot.get((direction, "best"), 0.0) + cnt * res[best] / 1e3
        rows.append(res)
        print(json.dumps(res))
summary = {f"{d}_{n}": round(v, 3) for (d, n), v in tot.items()}
print(json.dumps(summary))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"rows": rows, "per_step_ms": summary}, open("gpurun_out/conv_tile_sweep.json", "w"), indent=1)
