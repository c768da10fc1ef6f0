"""(development build, LK_LIB=laplace_amd/csrc/liblaplace_hip_dev.so) where the persistent window kernel's time goes: per
workgroup and tile the s_memtime stamps at K-loop start / K-loop end / epilogue end."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels
K = get_kernels(); dev = "cuda"
Co, Ci, H, N = 64, 64, 32, 1152
torch.manual_seed(0)
m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).to(dev)
g = K.split_f16x2((torch.randn(N, H, H, Co, device=dev) * 1e-3).contiguous())
add = K.split_f16x2((torch.randn(N, H, H, Ci, device=dev) * 1e-2).contiguous())
mask = (torch.rand(N // 9, H, H, Ci, device=dev) > 0.5).to(torch.uint8)
prep = cv.PreparedConv(m)
K.conv_config = 2 | (1 << 27)
for _ in range(3):
    cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16 * 3))()
K.lib.lk_winp_trace_read.restype = ctypes.c_int
K.lib.lk_winp_trace_read.argtypes = [ctypes.c_void_p]
assert K.lib.lk_winp_trace_read(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16, 3)[:512, :9].astype(np.float64)
t0 = t[:, 0, 0].min()
kl = t[:, :, 1] - t[:, :, 0]
ep = t[:, :, 2] - t[:, :, 1]
gap = t[:, 1:, 0] - t[:, :-1, 2]
tick = 1e-8  # s_memtime counts at 100 MHz on this chip
print(f"tiles per workgroup 9; s_memtime ticks -> us at {1 / tick / 1e6:.0f} MHz")
print("K loop per tile   (us): mean %.2f  p10 %.2f  p90 %.2f" % (kl.mean() * tick * 1e6, np.percentile(kl, 10) * tick * 1e6, np.percentile(kl, 90) * tick * 1e6))
print("epilogue per tile (us): mean %.2f  p10 %.2f  p90 %.2f" % (ep.mean() * tick * 1e6, np.percentile(ep, 10) * tick * 1e6, np.percentile(ep, 90) * tick * 1e6))
print("first K-loop start spread (us): %.2f;  last epilogue end - first start (us): %.2f" % ((t[:, 0, 0].max() - t0) * tick * 1e6, (t[:, 8, 2].max() - t0) * tick * 1e6))
for wg in (0, 1, 256, 257):
    print("wg", wg, "K:", np.round(kl[wg] * tick * 1e6, 1).tolist(), "E:", np.round(ep[wg] * tick * 1e6, 1).tolist(), "start", round((t[wg, 0, 0] - t0) * tick * 1e6, 1))
