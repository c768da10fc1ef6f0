"""One eigensolve of a realistic n x n KFAC-like factor (development tool for kernel traces)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import HipKernels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
K = HipKernels()
torch.manual_seed(0)
# decaying spectrum + low-rank bulk, like an accumulated A factor
X = torch.randn(3 * n // 2, n, device="cuda") * torch.logspace(0, -3, n, device="cuda")
A = (X.T @ X / X.shape[0]).contiguous()
for _ in range(2):
    w, Q, info = K.syevj_batched([A])[0]
torch.cuda.synchronize()
print("sweeps", int(info[1]), "info", int(info[0]))
