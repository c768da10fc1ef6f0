"""G-factor Grams of the c4 sweep's cotangents (lk_gram_tn_f16x2: split tensor [rows, C], rows = 9 seeds x 128 samples x pixels), per layer
shape: us per call (Gram + reduce), TB/s of operand bytes, TFLOP/s of fp16 MFMA work.  python tools/gram16_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = torch.device("cuda:0")
tot = 0.0
for (C, HW, n_layers) in ((64, 1024, 5), (128, 256, 5), (256, 64, 5), (512, 16, 5)):
    x = torch.randn(1152, HW, C, device=dev)
    xs = K.split_f16x2(x.view(1152, 1, HW, C))
    out = torch.zeros(C, C, device=dev)
    for _ in range(3):
        K.gram_tn_f16x2(xs, 0.5, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.gram_tn_f16x2(xs, 0.5, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    R = 1152 * HW
    tot += ms * n_layers
    print(f"C {C:4d} rows {R:8d}: {ms * 1e3:6.1f} us  {R * C * 4 / ms / 1e9:5.2f} TB/s of operand bytes  {R * C * (C + 1) * 3 / ms / 1e9:6.0f} TFLOP/s fp16")
print(f"20 layers of a c4 step: {tot:.3f} ms")
