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
Co, Ci, H, N = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 64, 32, 1152))]
print("shape Ci=%d Co=%d %dx%d N=%d" % (Ci, Co, H, H, N))
torch.manual_seed(0)
m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).to(dev)
g = K.split_f16x2((torch.randn(N, H, H, Co, device=dev) * 1e-3).contiguous())
add = K.split_f16x2((torch.randn(N, H, H, Ci, device=dev) * 1e-2).contiguous())
mask = (torch.rand(N // 9, H, H, Ci, device=dev) > 0.5).to(torch.uint8)
prep = cv.PreparedConv(m)
K.conv_config = int(sys.argv[6]) if len(sys.argv) > 6 else 2
for _ in range(3):
    cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16 * 3))()
K.lib.lk_winp_trace_read.restype = ctypes.c_int
K.lib.lk_winp_trace_read.argtypes = [ctypes.c_void_p]
assert K.lib.lk_winp_trace_read(buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16, 3)
NT = int(sys.argv[5]) if len(sys.argv) > 5 else 9
t = raw[:512, :NT].astype(np.float64)
valid = t[:, :, 2] > 0

hw = raw[:512, 15, 0].astype(np.int64); xcc = raw[:512, 15, 1].astype(np.int64) & 0xf
cu = (xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 7) << 4   # (XCC, SE_ID/SH, CU_ID): bits 8-11 CU, 12 SH, 13-15 SE
clk = 2.1e9
t0 = t[:, 0, 0].min()
kl = (t[:, :, 1] - t[:, :, 0]) / clk * 1e6
ep = (t[:, :, 2] - t[:, :, 1]) / clk * 1e6
if 0: print("K loop per tile (us at 2.1 GHz): mean %.2f p10 %.2f p90 %.2f;  epilogue: mean %.2f p10 %.2f p90 %.2f" % (kl.mean(), np.percentile(kl, 10), np.percentile(kl, 90), ep.mean(), np.percentile(ep, 10), np.percentile(ep, 90)))
kl = np.where(valid, kl, np.nan); ep = np.where(valid, ep, np.nan)
print("valid tiles per workgroup:", {int(n): int((valid.sum(1) == n).sum()) for n in set(valid.sum(1))})
print("K loop (valid) mean %.2f  epilogue mean %.2f" % (np.nanmean(kl), np.nanmean(ep)))
end = np.where(valid, t[:, :, 2], 0).max(1)
start = t[:, 0, 0]
print("launch span (us): %.1f ; workgroup start skew p50 %.1f max %.1f ; workgroup end: p10 %.1f p50 %.1f p90 %.1f max %.1f" % ((end.max() - t0) / clk * 1e6, np.percentile(start - t0, 50) / clk * 1e6, (start.max() - t0) / clk * 1e6, *[np.percentile(end - t0, q) / clk * 1e6 for q in (10, 50, 90)], (end.max() - t0) / clk * 1e6))
busy = np.nansum(kl, 1) + np.nansum(ep, 1)
print("sum of (K + epilogue) per workgroup: mean %.1f max %.1f us" % (busy.mean(), busy.max()))
w0 = raw[:512, 14, 0].astype(np.float64); w1 = raw[:512, 14, 1].astype(np.float64); m0_ = raw[:512, 15, 2].astype(np.float64); m1_ = raw[:512, 14, 2].astype(np.float64)
wt0 = w0.min()
print("wall clock (100 MHz): workgroup start p50 %.1f max %.1f us; end p10 %.1f p50 %.1f p90 %.1f max %.1f us; shader clock = %.2f GHz (s_memtime ticks per wall us, median)" % (np.percentile(w0 - wt0, 50) / 100, (w0.max() - wt0) / 100, *[np.percentile(w1 - wt0, q) / 100 for q in (10, 50, 90, 100)], np.median((m1_ - m0_) / ((w1 - w0) / 100)) / 1e3))
print("per workgroup: wall duration mean %.1f us; s_memtime duration / 2.1 GHz mean %.1f us; first K-loop stamp after start: %.1f us" % ((w1 - w0).mean() / 100, (m1_ - m0_).mean() / clk * 1e6, (t[:, 0, 0] - m0_).mean() / clk * 1e6))
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
        for i in range(NT):
            if not valid[w, i]: continue
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
e1.record(); torch.cuda.synchronize()
print("HIP events: %.1f us per launch (trace build)" % (e0.elapsed_time(e1) * 100))
order = np.argsort(w1)
print("latest workgroups (id, start us, end us, K per item, E per item):")
for b in order[-6:]:
    print("  wg %3d start %.1f end %.1f K %s E %s" % (b, (w0[b] - wt0) / 100, (w1[b] - wt0) / 100, np.round(kl[b] * 2.1 / 1.87, 1).tolist(), np.round(ep[b] * 2.1 / 1.87, 1).tolist()))
print("earliest finishing:")
for b in order[:4]:
    print("  wg %3d start %.1f end %.1f K %s E %s" % (b, (w0[b] - wt0) / 100, (w1[b] - wt0) / 100, np.round(kl[b] * 2.1 / 1.87, 1).tolist(), np.round(ep[b] * 2.1 / 1.87, 1).tolist()))
hist, edges = np.histogram((w1 - wt0) / 100, bins=12)
print("end-time histogram (us):", [(round(float(edges[i])), int(hist[i])) for i in range(len(hist))])
lo = (w0 - wt0) / 100 < 3
print("first workgroups of a CU (start < 3 us): %d, their end p50 %.1f max %.1f; the staggered ones: end p50 %.1f max %.1f" % (lo.sum(), np.percentile((w1[lo] - wt0) / 100, 50), ((w1[lo] - wt0) / 100).max(), np.percentile((w1[~lo] - wt0) / 100, 50), ((w1[~lo] - wt0) / 100).max()))
