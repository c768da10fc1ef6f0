"""Backward-data of the ResNet-18 3x3 convs at the sweep's batch (S*B = 1152): MIOpen solver choices (dev tool).
usage: conv_bwd_bench.py [channels_last]"""
import sys

import torch

cl = len(sys.argv) > 1 and sys.argv[1] == "channels_last"
dev = "cuda"
for cin, cout, hw, stride in [(64, 64, 32, 1), (128, 128, 16, 1), (256, 256, 8, 1), (512, 512, 4, 1), (64, 128, 32, 2),
                              (128, 256, 16, 2), (256, 512, 8, 2)]:
    oh = hw // stride
    w = torch.randn(cout, cin, 3, 3, device=dev)
    g = torch.randn(1152, cout, oh, oh, device=dev)
    dummy = torch.empty(1152, cin, hw, hw, device=dev)
    if cl:
        w = w.contiguous(memory_format=torch.channels_last)
        g = g.contiguous(memory_format=torch.channels_last)
        dummy = dummy.contiguous(memory_format=torch.channels_last)

    def run():
        return torch.ops.aten.convolution_backward(g, dummy, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1,
                                                   [True, False, False])[0]

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flops = 2.0 * 1152 * oh * oh * cin * cout * 9
    print(f"cin={cin:4d} cout={cout:4d} hw={hw:3d} s={stride} {'NHWC' if cl else 'NCHW'}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF effective", flush=True)
