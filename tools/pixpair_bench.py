"""Pixel-pair products of a 3x3 A factor over 8 stacked minibatches: one workgroup per pixel (13 shifts, `use_pixpair13`) against one per
(pixel, shift) block.  python tools/pixpair_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = torch.device("cuda:0")
for (B, C, H, W) in ((1024, 64, 32, 32), (1024, 128, 16, 16), (256, 64, 32, 32)):
    x = torch.randn(B, H, W, C, device=dev)
    xs = K.split_f16x2(x)
    plan = K.pixpair_plan(H, W, C, dev)
    res = {}
    for flag in (True, False):
        K.use_pixpair13 = flag
        blocks = torch.zeros(plan[0] * C * C, device=dev)
        for _ in range(3):
            K.pixpair_accumulate_split(xs, 0.5, blocks, plan)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.pixpair_accumulate_split(xs, 0.5, blocks, plan)
        e1.record()
        torch.cuda.synchronize()
        res[flag] = (e0.elapsed_time(e1) / 10, blocks.clone())
    flop = 13 * B * H * W * C * C * 2 * 3
    d = (res[True][1] - res[False][1]).abs().max().item() / res[False][1].abs().max().item()
    print(f"B {B} C {C} {H}x{W}: per pixel {res[True][0] * 1e3:.0f} us ({flop / res[True][0] / 1e9:.0f} TFLOP/s fp16), per block {res[False][0] * 1e3:.0f} us "
          f"({flop / res[False][0] / 1e9:.0f}); max rel diff {d:.2e}")
