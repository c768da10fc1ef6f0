"""Per-phase cycle counts of one Gram workgroup wave (development tool; needs the LK_GRAM_TRACE build
laplace_amd/csrc/liblaplace_hip_trace.so, see tools/build_trace.sh)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import _lib  # noqa: E402

path = os.path.join(os.path.dirname(_lib.LIB_PATH), "liblaplace_hip_trace.so")
K = _lib.HipKernels(_lib.load_library(path))
K.lib.lk_gram_trace_read.restype = ctypes.c_int
K.lib.lk_gram_trace_read.argtypes = [ctypes.c_void_p]
dev = "cuda"


def report(name):
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 5)()
    assert K.lib.lk_gram_trace_read(buf) == 0
    n = max(buf[4], 1)
    print(f"{name:28s} chunks={buf[4]:5d}  load-issue={buf[0] / n:7.0f}  mfma-block={buf[1] / n:7.0f}  "
          f"wait+lds-write={buf[2] / n:7.0f}  barrier={buf[3] / n:7.0f}  total/chunk={(buf[0] + buf[1] + buf[2] + buf[3]) / n:7.0f}", flush=True)


for rep in range(2):
    X = torch.randn(8192, 4608, device=dev)
    C = torch.zeros(4608, 4608, device=dev)
    K.gram_tn(X, 1.0, C)
    report("tn big 4608 K=8192")
    for name, cin, hw in (("conv l2 (n=1152)", 128, 16), ("conv l3 (n=2304)", 256, 8), ("conv l4 (n=4608)", 512, 4)):
        x = torch.randn(128, cin, hw, hw, device=dev)
        A = torch.zeros(cin * 9, cin * 9, device=dev)
        K.use_shiftcorr = False
        K.gram_conv(x, 3, 1, 1, 1, 1.0, A, upper_only=True, native=True)
        report(name)
    g = torch.randn(1152, 512, 16, device=dev)
    G = torch.zeros(512, 512, device=dev)
    K.gram_nt(g, 1.0, G, upper_only=True)
    report("nt G l4 (n=512, L=16)")
    g = torch.randn(1152, 128, 256, device=dev)
    G = torch.zeros(128, 128, device=dev)
    K.gram_nt(g, 1.0, G, upper_only=True)
    report("nt G l2 (n=128, L=256)")
