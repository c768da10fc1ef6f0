"""Workload for PMC passes: a few launches of the dominant Gram kernels at c4 shapes (development tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import HipKernels
K = HipKernels()
dev = "cuda"
B = 128
for name, cin, hw, s in [("l1", 64, 32, 1), ("l2", 128, 16, 1), ("l3", 256, 8, 1), ("l4", 512, 4, 1)]:
    x = torch.randn(B, cin, hw, hw, device=dev)
    n = cin * 9
    A = torch.zeros(n, n, device=dev)
    for _ in range(3):
        K.gram_conv(x, 3, s, 1, 1, 1.0, A, upper_only=True, native=True)
for n, L in [(64, 1024), (128, 256), (256, 64), (512, 16)]:
    g = [torch.randn(B, n, L, device=dev) for _ in range(10)]
    G = torch.zeros(n, n, device=dev)
    for _ in range(3):
        K.gram_nt(g, 1.0, G, upper_only=True)
torch.cuda.synchronize()
