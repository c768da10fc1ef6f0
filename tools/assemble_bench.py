"""Stand-alone timing of lk_conv3x3_pixpair_assemble2_f32 and of the ragged pixel-pair drain on the c4 layer geometries."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels
K = get_kernels()
dev = torch.device("cuda")
for Cin, H in ((64, 32), (128, 16), (256, 8), (512, 4)):
    plan = K.pixpair_plan(H, H, Cin, dev)
    nb = plan[0]
    b1 = torch.randn(nb * Cin * Cin, device=dev)
    b2 = torch.randn(nb * Cin * Cin, device=dev)
    A = torch.zeros(9 * Cin, 9 * Cin, device=dev)
    for two in (False, True):
        for _ in range(2):
            K.pixpair_assemble(b1, plan, H, H, Cin, 1.0, A, blocks2=b2 if two else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.pixpair_assemble(b1, plan, H, H, Cin, 1.0, A, blocks2=b2 if two else None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = b1.numel() * 4 * (2 if two else 1) / 1e9
        print(f"assemble Cin={Cin} {H}x{H} sets={2 if two else 1}: {ms * 1e3:7.1f} us  {gb / ms * 1e3:6.0f} GB/s ({gb:.2f} GB)")
    for nmb in (4, 8):
        x = torch.randn(nmb * 128, H, H, Cin, device=dev)
        for _ in range(2):
            K.pixpair_accumulate_split(K.split_f16x2(x), 0.5, b1, plan)
        torch.cuda.synchronize()
        xs = K.split_f16x2(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.pixpair_accumulate_split(xs, 0.5, b1, plan)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = (b1.numel() * 8 + x.numel() * 4) / 1e9
        print(f"drain    Cin={Cin} {H}x{H} minibatches={nmb}: {ms * 1e3:7.1f} us  {gb / ms * 1e3:6.0f} GB/s")
