"""The pixel-pair product (lk_conv3x3_pixpair_accumulate_f16x2) against the number of stacked minibatches: launch time per
c4 layer shape for 1 .. 16 minibatches of 128 images (development tool: what a partly filled group costs at the end of a fit)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = "cuda"


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, Cin in ((32, 64), (16, 128), (8, 256), (4, 512)):
    plan = K.pixpair_plan(H, H, Cin, torch.device(dev))
    blocks = torch.zeros(plan[0] * Cin * Cin, device=dev)
    row = [f"{Cin:4d} ch {H:2d}x{H:<2d} blocks {plan[0]:5d} ({blocks.numel() * 4 / 1e6:6.1f} MB) tiles {plan[1].shape[0]:6d}:"]
    for nmb in (1, 2, 4, 8, 16):
        x = torch.randn(nmb * 128, H, H, Cin, device=dev)
        xs = K.split_f16x2(x)
        us = timeit(lambda: K.pixpair_accumulate_split(xs, 1.0, blocks, plan))
        row.append(f"n={nmb}: {us:6.0f} us")
        del x, xs
    print("  ".join(row))
