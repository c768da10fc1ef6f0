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
K.conv_config = 2
for _ in range(3):
    cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16 * 3))()
K.lib.lk_winp_trace_read.restype = ctypes.c_int
K.lib.lk_winp_trace_read.argtypes = [ctypes.c_void_p]
assert K.lib.lk_winp_trace_read(buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16, 3)
t = raw[:512, :9].astype(np.float64)
hw = raw[:512, 15, 0].astype(np.int64); xcc = raw[:512, 15, 1].astype(np.int64) & 0xf
cu = (xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 7) << 4   # (XCC, SE_ID/SH, CU_ID): bits 8-11 CU, 12 SH, 13-15 SE
clk = 2.1e9
t0 = t[:, 0, 0].min()
kl = (t[:, :, 1] - t[:, :, 0]) / clk * 1e6
ep = (t[:, :, 2] - t[:, :, 1]) / clk * 1e6
print("K loop per tile (us at 2.1 GHz): mean %.2f p10 %.2f p90 %.2f;  epilogue: mean %.2f p10 %.2f p90 %.2f" % (kl.mean(), np.percentile(kl, 10), np.percentile(kl, 90), ep.mean(), np.percentile(ep, 10), np.percentile(ep, 90)))
print("launch span (us): %.1f" % ((t[:, 8, 2].max() - t0) / clk * 1e6))
groups = {}
for b in range(512):
    groups.setdefault(int(cu[b]), []).append(b)
sizes = sorted(len(v) for v in groups.values())
print("workgroups per (xcc, se, cu) id:", {n: sizes.count(n) for n in set(sizes)}, " ids:", len(groups))
# overlap of the two workgroups of a CU: fraction of the time both are in their K loops / both in epilogues
both_k = both_e = mixed = tot = 0.0
for ids in groups.values():
    if len(ids) != 2:
        continue
    a, b = ids
    ev = []
    for w in (a, b):
        for i in range(9):
            ev += [(t[w, i, 0], w, 'K'), (t[w, i, 1], w, 'E'), (t[w, i, 2], w, 'I')]
    ev.sort()
    state = {a: 'I', b: 'I'}
    last = ev[0][0]
    for tm, w, st in ev:
        d = tm - last
        sa, sb = state[a], state[b]
        if sa == 'K' and sb == 'K': both_k += d
        elif sa == 'E' and sb == 'E': both_e += d
        elif 'I' not in (sa, sb): mixed += d
        tot += d
        state[w] = st; last = tm
if tot:
    print("per CU with two workgroups: both in K loop %.0f %%, both in epilogue %.0f %%, one each %.0f %%" % (100 * both_k / tot, 100 * both_e / tot, 100 * mixed / tot))
for wg in (0, 1):
    print("wg", wg, "K:", np.round(kl[wg], 1).tolist(), "E:", np.round(ep[wg], 1).tolist())
