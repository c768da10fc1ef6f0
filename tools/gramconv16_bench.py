"""A factors of the strided / 1x1 convolutions of c4: the exact-fp32 MFMA kernel (lk_gram_conv_nhwc_f32) against the split-fp16
Gram engine on the patch matrix (development tool: is moving them onto gram_tn_f16x2 worth an im2col + split kernel?)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels, keep_layout

K = get_kernels()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, Cin, H, k, s, p in (("layer2.0 conv1", 64, 32, 3, 2, 1), ("layer3.0 conv1", 128, 16, 3, 2, 1), ("layer4.0 conv1", 256, 8, 3, 2, 1),
                              ("layer2.0 shortcut", 64, 32, 1, 2, 0), ("layer3.0 shortcut", 128, 16, 1, 2, 0), ("layer4.0 shortcut", 256, 8, 1, 2, 0),
                              ("conv1 (stem)", 3, 32, 3, 1, 1)):
    B = 128
    x = torch.randn(B, Cin, H, H, device="cuda").relu_().contiguous(memory_format=torch.channels_last)
    n = Cin * k * k
    A = torch.zeros(n, n, device="cuda")
    t32 = timeit(lambda: K.gram_conv(keep_layout(x), (k, k), (s, s), (p, p), (1, 1), 1e-3, A, upper_only=True, native=True))
    line = "%-18s n = %4d  fp32-MFMA gram_conv %6.1f us" % (name, n, t32)
    Kp = n if (n == 64 or n % 128 == 0) else (n + 127) // 128 * 128
    if Cin >= 8:
        cols = F.unfold(x, k, 1, p, s)                      # [B, Cin k k, L] in (ci, kh, kw) order: the order does not matter for timing
        rows = cols.transpose(1, 2).reshape(-1, n)
        if Kp != n:
            rows = torch.cat([rows, rows.new_zeros(rows.shape[0], Kp - n)], 1)
        rows = rows.contiguous()
        A2 = torch.zeros(Kp, Kp, device="cuda")
        t_split = timeit(lambda: K.split_f16x2(rows))
        sp = K.split_f16x2(rows)
        t16 = timeit(lambda: K.gram_tn_f16x2(sp, 1e-3, A2))
        line += "   split-fp16 Gram of the [%d, %d] patch matrix %6.1f us (+ its split %5.1f us)" % (rows.shape[0], Kp, t16, t_split)
    print(line, flush=True)
