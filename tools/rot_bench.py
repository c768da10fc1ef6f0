"""The Kron predictive's eigenbasis rotation of the unfolded inputs (a 3x3 convolution Cin -> 9 Cin that emits chunk-major split
planes: lk_conv_nhwc_f16x2_planes) per c4 layer shape, by tile shape (conv_config bits 12..14; 0 = the occupancy rule).
python tools/rot_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = torch.device("cuda:0")
B = 128
tot = {}
seen = {}
for (C, H, n_layers) in ((64, 32, 4), (128, 16, 3), (256, 8, 3), (512, 4, 3)):
    m = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
    a = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    Q = torch.linalg.qr(torch.randn(9 * C, 9 * C, device=dev))[0]
    filt = Q.t().reshape(9 * C, 3, 3, C).permute(0, 3, 1, 2).contiguous()
    flop = 2.0 * B * H * H * (9 * C) * (9 * C)
    line = f"C {C:4d} {H:2d}x{H:<2d}:"
    ref = None
    for tile in (0, 4, 5, 3, 2, 0):
        prev = K.conv_config
        K.conv_config = 2 | (tile << 12)
        try:
            for _ in range(3):
                out = cv.conv_forward_filters(m, a, filt, Q, planes=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = cv.conv_forward_filters(m, a, filt, Q, planes=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
        finally:
            K.conv_config = prev
        v = out.float()
        if ref is None:
            ref = v
        d = (v - ref).abs().max().item() / ref.abs().max().item()
        tot[tile] = ms * n_layers + (tot.get(tile, 0.0) if (C, tile) not in seen else tot[tile] - seen[(C, tile)]); seen[(C, tile)] = ms * n_layers
        line += f"  tile {tile}: {ms * 1e3:6.0f} us {flop / ms / 1e9:4.0f} TF (diff {d:.0e})"
    print(line)
print("per predictive call (13 layers):", {k: round(v, 3) for k, v in tot.items()})
